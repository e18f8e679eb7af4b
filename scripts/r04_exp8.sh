mkdir -p gpurun_out; L=gpurun_out/r04_exp8.log; rm -f $L
one() {  lbl=$1; shift
  env "$@" timeout -k 10 200 python bench.py --steps 400 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('%-34s step %7.1f us  warp %6.1f us  gram %6.1f us'%('$lbl', d['ms_per_step']*1e3, r['kernel_ms']*1e3, r['accumulate_kernel_ms']*1e3))" | tee -a $L; }
one default X=1
for v in $(ls build_variants/*.so | grep -v diag); do one $v EMBA_LIB=$PWD/$v; done
one default X=1
