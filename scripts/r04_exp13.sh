# list-driven gather beside the record stream (4 gather waves per Gram block): where does it beat the sweeping active-write launch now?
mkdir -p gpurun_out; L=gpurun_out/r04_exp13.log; rm -f $L
run() { # label events pano_h K steps extra-args env...
  lbl=$1; n=$2; ph=$3; k=$4; st=$5; extra=$6; shift 6
  env "$@" timeout -k 10 400 python bench.py --steps $st --warmup 2 --no-cpu-baseline --events-per-gpu $n --pano-h $ph --knots $k $extra 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); r=d['roofline']; c=d['config']; s=c['setup']
print('%-30s N=%9d pano_h=%4d K=%3d: %7.3f G ev/s  step %9.1f us  warp %8.1f us  gram %8.1f us  frac %.3f | %s'%('$lbl', c['events_per_rank'], $ph, $k, d['value']/1e9, d['ms_per_step']*1e3, r['kernel_ms']*1e3, r['accumulate_kernel_ms']*1e3, r['frac'], 'tile' if s['tile_order'] else 'pixel'))" | tee -a $L
}
for g in 2 3; do
run "1.5M gather=$g" 1500000 1024 21 100 "" EMBA_STEP_GATHER=$g
run "2M gather=$g" 2000000 1024 21 60 "" EMBA_STEP_GATHER=$g
run "3M gather=$g" 3000000 1024 21 40 "" EMBA_STEP_GATHER=$g
run "5M K97 gather=$g" 5000000 1024 97 20 "" EMBA_STEP_GATHER=$g
run "10M K97 gather=$g" 10000000 1024 97 10 "" EMBA_STEP_GATHER=$g
run "10M 640x480 gather=$g" 10000000 1024 97 10 "--sensor 640x480" EMBA_STEP_GATHER=$g
run "10M 2048 K256 gather=$g" 10000000 2048 256 10 "" EMBA_STEP_GATHER=$g
run "40M 2048 K97 gather=$g" 40000000 2048 97 5 "" EMBA_STEP_GATHER=$g
done
