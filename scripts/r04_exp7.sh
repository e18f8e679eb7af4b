mkdir -p gpurun_out; L=gpurun_out/r04_exp7.log; rm -f $L
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "baseline_size or resident_step or normal_equations or irls or randomised or other_configurations or panorama_border" > gpurun_out/r04_tests2.log 2>&1; rc=$?
tail -3 gpurun_out/r04_tests2.log | tee -a $L
[ $rc -ne 0 ] && { grep -E "Error|assert|FAILED|error" gpurun_out/r04_tests2.log | head -30; exit $rc; }
TAG=r04j STEPS=300 bash scripts/quick_trace.sh 2>&1 | head -6 | tee -a $L
for i in 1 2; do timeout -k 10 200 python bench.py --steps 400 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('step %7.1f us  warp %6.1f us  gram %6.1f us'%(d['ms_per_step']*1e3, r['kernel_ms']*1e3, r['accumulate_kernel_ms']*1e3))" | tee -a $L; done
