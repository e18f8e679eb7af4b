# GPU round trip: parity tests, then a short bench (used during development; the driver runs pytest/bench itself)
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1; rc=$?
tail -4 gpurun_out/gpu_tests.log
[ $rc -ne 0 ] && { grep -E "Error|assert|FAILED" gpurun_out/gpu_tests.log | head -30; exit $rc; }
timeout -k 10 300 python bench.py --steps 100 --warmup 5 --no-cpu-baseline > gpurun_out/bench_quick.json 2>gpurun_out/bench_quick.err || { tail -5 gpurun_out/bench_quick.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_quick.json').readline()); r=d['roofline']
print('value %.3f G ev/s  step %.1f us  warp %.1f us  gram %.1f us  frac %.3f path_frac %.3f'%(d['value']/1e9, d['ms_per_step']*1e3, r['kernel_ms']*1e3, r['accumulate_kernel_ms']*1e3, r['frac'], r['path_frac']))
PY
