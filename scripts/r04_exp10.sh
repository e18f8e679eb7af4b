mkdir -p gpurun_out; L=gpurun_out/r04_exp10.log; rm -f $L
run() { # label events pano_h K steps extra-args env...
  lbl=$1; n=$2; ph=$3; k=$4; st=$5; extra=$6; shift 6
  env "$@" timeout -k 10 400 python bench.py --steps $st --warmup 2 --no-cpu-baseline --events-per-gpu $n --pano-h $ph --knots $k $extra 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); r=d['roofline']; c=d['config']; s=c['setup']
print('%-34s N=%9d pano_h=%4d K=%3d: %7.3f G ev/s  step %9.1f us  warp %8.1f us  gram %8.1f us  frac %.3f | %s'%('$lbl', c['events_per_rank'], $ph, $k, d['value']/1e9, d['ms_per_step']*1e3, r['kernel_ms']*1e3, r['accumulate_kernel_ms']*1e3, r['frac'], 'tile' if s['tile_order'] else 'pixel'))" | tee -a $L
}
for o in pixel tile; do
run "2M $o" 2000000 1024 21 50 "" EMBA_ORDER=$o
run "2.5M $o" 2500000 1024 21 50 "" EMBA_ORDER=$o
run "3M $o" 3000000 1024 21 50 "" EMBA_ORDER=$o
run "5M K97 $o" 5000000 1024 97 20 "" EMBA_ORDER=$o
run "10M K97 $o" 10000000 1024 97 10 "" EMBA_ORDER=$o
run "shard 5M of 40M 640x480 $o" 5000000 1024 97 20 "--sensor 640x480 --shard-of 8 --shard-rank 3 --yaw-rate 0.1" EMBA_ORDER=$o
run "shard 12.5M of 100M $o" 12500000 2048 256 8 "--shard-of 8 --shard-rank 3" EMBA_ORDER=$o
done
