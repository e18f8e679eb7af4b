# one rank's shard of the 8-GPU configurations on one GPU (bench.py --shard-of 8), pixel vs tile order (profiles/r03_shard_sweep.txt)
mkdir -p gpurun_out; rm -f gpurun_out/shard_sweep.log
run() { for o in ${ORDERS:-auto}; do EMBA_ORDER=$o timeout -k 10 500 python bench.py --steps ${4:-10} --warmup 2 --no-cpu-baseline --events-per-gpu $1 --pano-h $2 --knots $3 --shard-of 8 --shard-rank ${6:-3} ${5:-} 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); r=d['roofline']; c=d['config']; s=c['setup']
print('shard of 8 x %9d pano_h=%4d K=%3d %-5s: %7.3f G ev/s  step %9.1f us  warp %8.1f us  gram %8.1f us  frac %.3f  inl %d P %d | %s entries %d chunks %d %s'%($1, $2, $3, '$o', d['value']/1e9, d['ms_per_step']*1e3, r['kernel_ms']*1e3, r['accumulate_kernel_ms']*1e3, r['frac'], c['inliers_rank0'], c['active_pixels'], 'tile' if s['tile_order'] else 'pixel', s['entries'], s['chunks'], '${5:-}'))" >> gpurun_out/shard_sweep.log; done; }
ORDERS="pixel tile auto"
run 1000000 1024 21 100
run 5000000 1024 97 10 "--sensor 640x480 --yaw-rate 0.1" 5
run 12500000 2048 256 6
cat gpurun_out/shard_sweep.log
