# development: time alternative builds of the library (build_variants/*.so, same ABI) against the default build
mkdir -p gpurun_out; rm -f gpurun_out/variants.log
for v in default $(ls build_variants/*.so 2>/dev/null); do
  [ "$v" = default ] && unset EMBA_LIB || export EMBA_LIB=$PWD/$v
  echo "== $v" >> gpurun_out/variants.log
  timeout -k 10 120 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('  step %.1f us  warp %.1f us  gram %.1f us'%(d['ms_per_step']*1e3, r['kernel_ms']*1e3, r['accumulate_kernel_ms']*1e3))" >> gpurun_out/variants.log
done
cat gpurun_out/variants.log
