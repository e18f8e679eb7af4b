mkdir -p gpurun_out && rm -f gpurun_out/r04_exp6.log
L=gpurun_out/r04_exp6.log
one() {  # label, env...
  lbl=$1; shift
  env "$@" timeout -k 10 200 python bench.py --steps ${STEPS:-400} --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('%-34s step %7.1f us  warp %6.1f us  gram %6.1f us'%('$lbl', d['ms_per_step']*1e3, r['kernel_ms']*1e3, r['accumulate_kernel_ms']*1e3))" | tee -a $L
}
one "default (one set)" X=1
one "two sets" EMBA_STEP_ONE_SET=0
one "one set, 100 steps" STEPS=100
one "two sets, 100 steps" EMBA_STEP_ONE_SET=0 STEPS=100
one "two sets, 20 steps" EMBA_STEP_ONE_SET=0 STEPS=20
