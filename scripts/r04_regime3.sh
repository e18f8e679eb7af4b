mkdir -p gpurun_out; L=gpurun_out/r04_regime3.log; rm -f $L
run() { # label events pano_h K steps extra-args env...
  lbl=$1; n=$2; ph=$3; k=$4; st=$5; extra=$6; shift 6
  env "$@" timeout -k 10 400 python bench.py --steps $st --warmup 2 --no-cpu-baseline --events-per-gpu $n --pano-h $ph --knots $k $extra 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); r=d['roofline']; c=d['config']; s=c['setup']
print('%-34s N=%9d pano_h=%4d K=%3d: %7.3f G ev/s  step %9.1f us  warp %8.1f us  gram %8.1f us  frac %.3f | %s entries %d chunks %d'%('$lbl', c['events_per_rank'], $ph, $k, d['value']/1e9, d['ms_per_step']*1e3, r['kernel_ms']*1e3, r['accumulate_kernel_ms']*1e3, r['frac'], 'tile' if s['tile_order'] else 'pixel', s['entries'], s['chunks']))" | tee -a $L
}
for n in 1000000 1200000 1500000; do
  run "$n compact" $n 1024 21 100 "" EMBA_GRAM=compact EMBA_ORDER=pixel
  run "$n stream" $n 1024 21 100 "" EMBA_GRAM=stream EMBA_ORDER=pixel
done
run "shard 1M of 8M compact" 1000000 1024 21 100 "--shard-of 8 --shard-rank 3" EMBA_GRAM=compact
run "shard 1M of 8M stream" 1000000 1024 21 100 "--shard-of 8 --shard-rank 3" EMBA_GRAM=stream
run "1M 512x1024 compact" 1000000 512 21 100 "" EMBA_GRAM=compact
run "1M 512x1024 stream" 1000000 512 21 100 "" EMBA_GRAM=stream
run "scene compact" 1000000 1024 21 100 "--data scene" EMBA_GRAM=compact
run "scene stream" 1000000 1024 21 100 "--data scene" EMBA_GRAM=stream
run "10M K97" 10000000 1024 97 10 "" X=1
run "10M K201 10s" 10000000 1024 201 10 "" X=1
run "40M" 40000000 2048 97 5 "" X=1
run "100M" 100000000 2048 256 3 "" X=1
