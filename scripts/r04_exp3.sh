# round 4, experiment 3: anatomy of the compact Gram kernel (diag build: s_memtime stamps + ablations)
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
one "diag default" EMBA_LIB=$D EMBA_GRAM_TRACE=1
one "diag stream form" EMBA_LIB=$D EMBA_GRAM=stream
for a in 32 64 96 256; do one "compact ablate $a" EMBA_LIB=$D EMBA_ABLATE=$a EMBA_GRAM_TRACE=1; done
