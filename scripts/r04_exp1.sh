# round 4, experiment 1 (1 M-event BASELINE workload): -m gpu tests, A/B of the step variants, per-kernel trace, warp-kernel ablation, SQ counters
mkdir -p gpurun_out && rm -f gpurun_out/r04_exp1.log
L=gpurun_out/r04_exp1.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r04_tests.log | tee -a $L
[ $rc -ne 0 ] && { grep -E "Error|assert|FAILED" gpurun_out/r04_tests.log | head -20; exit $rc; }
one() {  # label, env...
  lbl=$1; shift
  env "$@" timeout -k 10 200 python bench.py --steps ${STEPS:-400} --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('%-34s step %7.1f us  warp %6.1f us  gram %6.1f us'%('$lbl', d['ms_per_step']*1e3, r['kernel_ms']*1e3, r['accumulate_kernel_ms']*1e3))" | tee -a $L
}
one "default(fast+compact U4)" X=1
one "EMBA_STEP_FAST=0" EMBA_STEP_FAST=0
one "EMBA_GRAM=stream" EMBA_GRAM=stream
one "fast=0,stream (round 3)" EMBA_STEP_FAST=0 EMBA_GRAM=stream
for U in 2 6 8; do one "compact U$U" EMBA_LIB=$PWD/build_variants/compact_u$U.so; done
one "default again" X=1
echo "--- warp kernel ablation (diag build; bits: 1 no marker store, 2 no record store, 8 no pixacc atomics, 4 no texel gather)" | tee -a $L
for a in 0 1 2 8 3 9 10 11 15; do one "ablate $a" EMBA_LIB=$PWD/build_variants/diag.so EMBA_ABLATE=$a STEPS=200; done
TAG=r04a STEPS=300 bash scripts/quick_trace.sh 2>&1 | tee -a $L
python scripts/step_timeline.py gpurun_out/trace_r04a/trace 2>&1 | tee -a $L
PMC="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" TAG=r04_sq bash scripts/pmc.sh 2>&1 | grep -E "warp_residual|gram|post_warp|active_write|prep_pose" | tee -a $L
PMC="SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM" TAG=r04_sq2 bash scripts/pmc.sh 2>&1 | grep -E "warp_residual|gram|post_warp|active_write|prep_pose" | tee -a $L
