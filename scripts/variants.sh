# development: time alternative builds of the library (build_variants/*.so, same ABI) against the default build
#   ARGS="--events-per-gpu 10000000 --knots 97" ORD=tile bash scripts/variants.sh
mkdir -p gpurun_out; rm -f gpurun_out/variants.log
for v in default $(ls build_variants/*.so 2>/dev/null); do
  [ "$v" = default ] && unset EMBA_LIB || export EMBA_LIB=$PWD/$v
  EMBA_ORDER=${ORD:-auto} timeout -k 10 200 python bench.py --steps ${STEPS:-10} --warmup 3 --no-cpu-baseline $ARGS 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('%-36s step %9.1f us  warp %9.1f us  gram %8.1f us'%('$v', d['ms_per_step']*1e3, r['kernel_ms']*1e3, r['accumulate_kernel_ms']*1e3))" >> gpurun_out/variants.log
done
cat gpurun_out/variants.log
