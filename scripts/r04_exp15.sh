# gather waves per Gram block (1 / 2 / 4) with the list-driven gather forced everywhere (EMBA_STEP_GATHER=3) against the default policy
mkdir -p gpurun_out; L=gpurun_out/r04_exp15.log; rm -f $L
run() { # label events pano_h K steps extra-args env...
  lbl=$1; n=$2; ph=$3; k=$4; st=$5; extra=$6; shift 6
  env "$@" timeout -k 10 400 python bench.py --steps $st --warmup 2 --no-cpu-baseline --events-per-gpu $n --pano-h $ph --knots $k $extra 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); r=d['roofline']; c=d['config']; s=c['setup']
print('%-26s N=%9d pano_h=%4d K=%3d: %7.3f G ev/s  step %9.1f us  warp %8.1f us  gram %8.1f us | %s'%('$lbl', c['events_per_rank'], $ph, $k, d['value']/1e9, d['ms_per_step']*1e3, r['kernel_ms']*1e3, r['accumulate_kernel_ms']*1e3, 'tile' if s['tile_order'] else 'pixel'))" | tee -a $L
}
sizes() { # env...
  run "$1 1M" 1000000 1024 21 300 "" "${@:2}"
  run "$1 3M" 3000000 1024 21 40 "" "${@:2}"
  run "$1 5M K97" 5000000 1024 97 20 "" "${@:2}"
  run "$1 10M K97" 10000000 1024 97 10 "" "${@:2}"
  run "$1 10M 640x480" 10000000 1024 97 10 "--sensor 640x480" "${@:2}"
  run "$1 10M 2048 K256" 10000000 2048 256 10 "" "${@:2}"
  run "$1 40M 2048 K97" 40000000 2048 97 5 "" "${@:2}"
}
sizes "sweep (no lists)" EMBA_STEP_GATHER=0
sizes "gw1" EMBA_STEP_GATHER=3 EMBA_GATHER_WAVES=1
sizes "gw2" EMBA_STEP_GATHER=3 EMBA_GATHER_WAVES=2
sizes "gw4" EMBA_STEP_GATHER=3 EMBA_GATHER_WAVES=4
