# evidence (round 6): pixel vs tile order and the tile reserve inside an LM loop at config 2 shape, plus the steady step under both orders (profiles/r06_lm_order_ab.txt)
mkdir -p gpurun_out
for o in "" "order=1" "order=2,tile_reserve=4" "order=2,tile_reserve=5,tile_shape=0"; do
  echo "=== EMBA_OPTS=$o"
  EMBA_OPTS=$o timeout -k 10 300 python scripts/lm_timing.py 2>&1 | grep -v amdgpu.ids | grep -A6 "window 2" | cut -c1-600
done
for o in "order=1" "order=2"; do
  timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-with-ep --long-steps 0 --events-per-gpu 10000000 --knots 201 --opt $o 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); r=d['roofline']; c=d['config']
print('config-2 shape steady step $o: step %.1f us warp %.1f gram %.1f  %s'%(d['ms_per_step']*1e3, r['kernel_ms_raw']*1e3, r['accumulate_kernel_ms']*1e3, c['setup']))"
done
