# development: fused Gram sums (tile order) on / off over the sizes where the tile order is chosen
mkdir -p gpurun_out; rm -f gpurun_out/fuse_ab.log
run() { for f in 1 0; do EMBA_FUSE_GRAM=$f timeout -k 10 400 python bench.py --steps ${4:-10} --warmup 2 --no-cpu-baseline --events-per-gpu $1 --pano-h $2 --knots $3 ${5:-} 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); r=d['roofline']; c=d['config']; s=c['setup']
print('N=%9d pano_h=%4d K=%3d fuse=%s: %7.3f G ev/s  step %9.1f us  warp %8.1f us  gram/correct %8.1f us  frac %.3f  inl %d P %d | %s'%(c['total_events'], $2, $3, '$f', d['value']/1e9, d['ms_per_step']*1e3, r['kernel_ms']*1e3, r['accumulate_kernel_ms']*1e3, r['frac'], c['inliers_rank0'], c['active_pixels'], 'tile' if s['tile_order'] else 'pixel'))" >> gpurun_out/fuse_ab.log; done; }
run 3000000 1024 21 20
run 5000000 1024 97 10
run 10000000 1024 97 10
run 10000000 1024 97 10 "--sensor 640x480"
run 40000000 2048 97 5
run 100000000 2048 256 3
cat gpurun_out/fuse_ab.log
