# Gram kernel, GATHER form: the gather as the work of the block's first GRAM_GATHER_WAVES waves beside the record stream (default 4, GATHER_Q 4 passes in flight)
# vs all 16 waves first (old_head: the build of the commit before).  Variants are built by hand into build_variants/ (hipcc -DGRAM_GATHER_WAVES=.. -DGATHER_Q=..).
mkdir -p gpurun_out; L=gpurun_out/r04_exp12.log; rm -f $L
run() { lbl=$1; lib=$2; extra=$3
  [ -f $lib ] || return 0
  EMBA_LIB=$lib timeout -k 10 400 python bench.py --steps 300 --no-cpu-baseline $extra 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('%-28s step %7.1f us  warp %6.1f us  gram %6.1f us'%('$lbl', d['ms_per_step']*1e3, r['kernel_ms']*1e3, r['accumulate_kernel_ms']*1e3))" | tee -a $L
}
for rep in 1 2; do
run "old head (16 waves first)" $PWD/build_variants/old_head.so ""
run "gather waves 2" $PWD/build_variants/gw2.so ""
run "gather waves 3" $PWD/build_variants/gw3.so ""
run "gather waves 4 (default)" $PWD/emba_amd/libemba_hip.so ""
run "gather waves 6" $PWD/build_variants/gw6.so ""
for v in gw3q8 gw4q8 gw5q4 gw4q2; do run "$v" $PWD/build_variants/$v.so ""; done
done
run "scene old" $PWD/build_variants/old_head.so "--data scene"
run "scene gw4" $PWD/emba_amd/libemba_hip.so "--data scene"
run "shard old" $PWD/build_variants/old_head.so "--shard-of 8 --shard-rank 3"
run "shard gw4" $PWD/emba_amd/libemba_hip.so "--shard-of 8 --shard-rank 3"
