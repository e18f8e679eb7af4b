# round 4: the 2-5 M-event regime and the shard workloads — chunk order (longest first vs bin order), order policy at 10 M / 2048x4096
mkdir -p gpurun_out; L=gpurun_out/r04_regime.log; rm -f $L
run() { # label events pano_h K steps extra-args env...
  lbl=$1; n=$2; ph=$3; k=$4; st=$5; extra=$6; shift 6
  env "$@" timeout -k 10 400 python bench.py --steps $st --warmup 2 --no-cpu-baseline --events-per-gpu $n --pano-h $ph --knots $k $extra 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); r=d['roofline']; c=d['config']; s=c['setup']
print('%-34s N=%9d pano_h=%4d K=%3d: %7.3f G ev/s  step %9.1f us  warp %8.1f us  gram %8.1f us  frac %.3f | %s entries %d chunks %d'%('$lbl', c['events_per_rank'], $ph, $k, d['value']/1e9, d['ms_per_step']*1e3, r['kernel_ms']*1e3, r['accumulate_kernel_ms']*1e3, r['frac'], 'tile' if s['tile_order'] else 'pixel', s['entries'], s['chunks']))" | tee -a $L
}
run "2M lpt" 2000000 1024 21 30 "" X=1
run "2M bin" 2000000 1024 21 30 "" EMBA_CHUNK_ORDER=bin
run "3M lpt" 3000000 1024 21 30 "" X=1
run "3M bin" 3000000 1024 21 30 "" EMBA_CHUNK_ORDER=bin
run "5M K97 lpt" 5000000 1024 97 20 "" X=1
run "5M K97 bin" 5000000 1024 97 20 "" EMBA_CHUNK_ORDER=bin
run "shard 5M of 40M 640x480 lpt" 5000000 1024 97 20 "--sensor 640x480 --shard-of 8 --shard-rank 3 --yaw-rate 0.1" X=1
run "shard 5M of 40M 640x480 bin" 5000000 1024 97 20 "--sensor 640x480 --shard-of 8 --shard-rank 3 --yaw-rate 0.1" EMBA_CHUNK_ORDER=bin
run "10M 2048x4096 K256 auto" 10000000 2048 256 10 "" X=1
run "10M 2048x4096 K256 tile" 10000000 2048 256 10 "" EMBA_ORDER=tile
run "10M 2048x4096 K256 pixel" 10000000 2048 256 10 "" EMBA_ORDER=pixel
run "10M K97 lpt" 10000000 1024 97 10 "" X=1
run "10M K97 bin" 10000000 1024 97 10 "" EMBA_CHUNK_ORDER=bin
run "shard 12.5M of 100M lpt" 12500000 2048 256 8 "--shard-of 8 --shard-rank 3" X=1
run "shard 12.5M of 100M bin" 12500000 2048 256 8 "--shard-of 8 --shard-rank 3" EMBA_CHUNK_ORDER=bin
