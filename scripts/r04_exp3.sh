# round 4: anatomy of the compact Gram kernel (diag build: s_memtime stamps per wave)
mkdir -p gpurun_out && rm -f gpurun_out/r04_exp3.log
L=gpurun_out/r04_exp3.log
one() {  # label, env...
  lbl=$1; shift
  env "$@" timeout -k 10 200 python bench.py --steps ${STEPS:-300} --warmup 3 --no-cpu-baseline 2>gpurun_out/r04_exp3.err | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('%-34s step %7.1f us  warp %6.1f us  gram %6.1f us'%('$lbl', d['ms_per_step']*1e3, r['kernel_ms']*1e3, r['accumulate_kernel_ms']*1e3))" | tee -a $L
  grep "gram trace" gpurun_out/r04_exp3.err | tee -a $L
}
D=$PWD/build_variants/diag.so
one "diag gather in head" EMBA_LIB=$D EMBA_GRAM_TRACE=1
one "diag gather standalone" EMBA_LIB=$D EMBA_GRAM_TRACE=1 EMBA_STEP_GATHER=1
