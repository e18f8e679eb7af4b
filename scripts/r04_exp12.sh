# Gram kernel, GATHER form: the gather as the work of the block's first GRAM_GATHER_WAVES waves beside the record stream vs all 16 waves first (old_head)
mkdir -p gpurun_out; L=gpurun_out/r04_exp12.log
run() { lbl=$1; lib=$2; extra=$3
  EMBA_LIB=$lib timeout -k 10 400 python bench.py --steps 300 --no-cpu-baseline $extra 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('%-28s step %7.1f us  warp %6.1f us  gram %6.1f us'%('$lbl', d['ms_per_step']*1e3, r['kernel_ms']*1e3, r['accumulate_kernel_ms']*1e3))" | tee -a $L
}
for rep in 1 2; do
run "gw4 q4 (default)" $PWD/emba_amd/libemba_hip.so ""
for v in gw3q8 gw4q8 gw5q4 gw4q2; do run "$v" $PWD/build_variants/$v.so ""; done
done
