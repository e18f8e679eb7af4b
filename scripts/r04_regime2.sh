mkdir -p gpurun_out; L=gpurun_out/r04_regime2.log; rm -f $L
run() { # label events pano_h K steps extra-args env...
  lbl=$1; n=$2; ph=$3; k=$4; st=$5; extra=$6; shift 6
  env "$@" timeout -k 10 400 python bench.py --steps $st --warmup 2 --no-cpu-baseline --events-per-gpu $n --pano-h $ph --knots $k $extra 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); r=d['roofline']; c=d['config']; s=c['setup']
print('%-34s N=%9d pano_h=%4d K=%3d: %7.3f G ev/s  step %9.1f us  warp %8.1f us  gram %8.1f us  frac %.3f | %s entries %d chunks %d'%('$lbl', c['events_per_rank'], $ph, $k, d['value']/1e9, d['ms_per_step']*1e3, r['kernel_ms']*1e3, r['accumulate_kernel_ms']*1e3, r['frac'], 'tile' if s['tile_order'] else 'pixel', s['entries'], s['chunks']))" | tee -a $L
}
for v in default tmg8 tmg12; do
  [ $v = default ] && E="X=1" || E="EMBA_LIB=$PWD/build_variants/$v.so"
  run "2M $v" 2000000 1024 21 30 "" $E
  run "3M $v" 3000000 1024 21 30 "" $E
  run "5M K97 $v" 5000000 1024 97 20 "" $E
  run "shard 5M of 40M $v" 5000000 1024 97 20 "--sensor 640x480 --shard-of 8 --shard-rank 3 --yaw-rate 0.1" $E
done
run "1.5M pixel" 1500000 1024 21 30 "" EMBA_ORDER=pixel
run "1.5M tile" 1500000 1024 21 30 "" EMBA_ORDER=tile
run "2M pixel" 2000000 1024 21 30 "" EMBA_ORDER=pixel
