# round 4, experiment 2: fused active-write in the Gram head — tests, A/B, trace
mkdir -p gpurun_out && rm -f gpurun_out/r04_exp2.log
L=gpurun_out/r04_exp2.log
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "resident_step or single_call_step or declared_cost or lm_loop or edge_cases" > gpurun_out/r04_tests2.log 2>&1; rc=$?
tail -3 gpurun_out/r04_tests2.log | tee -a $L
[ $rc -ne 0 ] && { grep -E "Error|assert|FAILED|error" gpurun_out/r04_tests2.log | head -30; exit $rc; }
one() {  # label, env...
  lbl=$1; shift
  env "$@" timeout -k 10 200 python bench.py --steps ${STEPS:-400} --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('%-34s step %7.1f us  warp %6.1f us  gram %6.1f us'%('$lbl', d['ms_per_step']*1e3, r['kernel_ms']*1e3, r['accumulate_kernel_ms']*1e3))" | tee -a $L
}
one "default (AW on side stream)" X=1
one "EMBA_STEP_SIDE=0" EMBA_STEP_SIDE=0
one "SIDE=0 STEP_FAST=0" EMBA_STEP_SIDE=0 EMBA_STEP_FAST=0
one "default again" X=1
TAG=r04c STEPS=300 bash scripts/quick_trace.sh 2>&1 | tee -a $L
python scripts/step_timeline.py gpurun_out/trace_r04c/trace 2>&1 | tee -a $L
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r04_tests.log | tee -a $L
[ $rc -ne 0 ] && { grep -E "Error|assert|FAILED" gpurun_out/r04_tests.log | head -20; exit $rc; }
exit 0
