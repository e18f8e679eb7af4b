# single-GPU size sweep of the hot path (SURVEY §8d scaling set); prints one line per configuration.  ORDERS="auto pixel tile"
mkdir -p gpurun_out; rm -f gpurun_out/scaling.log
run() { for o in ${ORDERS:-auto}; do case $o in pixel) oo=1;; tile) oo=2;; *) oo=0;; esac; timeout -k 10 400 python bench.py --opt order=$oo --steps ${4:-10} --warmup 2 --no-cpu-baseline --events-per-gpu $1 --pano-h $2 --knots $3 ${5:-} 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); r=d['roofline']; c=d['config']; s=c['setup']
print('N=%9d pano_h=%4d K=%3d %-5s: %7.3f G ev/s  step %9.1f us  warp %8.1f us  gram %8.1f us  frac %.3f  inl %d P %d | %s entries %d chunks %d set_events %.1f ms prepare %.1f ms'%(c['total_events'], $2, $3, '$o', d['value']/1e9, d['ms_per_step']*1e3, r['kernel_ms']*1e3, r['accumulate_kernel_ms']*1e3, r['frac'], c['inliers_rank0'], c['active_pixels'], 'tile' if s['tile_order'] else 'pixel', s['entries'], s['chunks'], s['set_events_ms'], s['prepare_ms']))" >> gpurun_out/scaling.log; done; }
run 1000000 512 21 20
run 1000000 1024 21 20
run 10000000 1024 201 10      # config 2's shape (shapes.launch: 10 s at dt = 0.05 s, K = 201) at the BASELINE event rate
run 3000000 1024 21 20
run 5000000 1024 97 10
run 10000000 1024 97 10
run 10000000 1024 97 10 "--sensor 640x480"
run 10000000 2048 256 10
run 40000000 2048 97 5
run 100000000 2048 256 3
cat gpurun_out/scaling.log
