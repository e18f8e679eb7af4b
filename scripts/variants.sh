# development: time alternative builds of the library (build_variants/*.so, same ABI) against the default build, interleaved on one box
#   ARGS="--events-per-gpu 10000000 --knots 97" REPS=2 bash scripts/variants.sh
mkdir -p gpurun_out; rm -f gpurun_out/variants.log
for rep in $(seq 1 ${REPS:-2}); do
for v in default $(ls build_variants/*.so 2>/dev/null); do
  [ "$v" = default ] && unset EMBA_LIB || export EMBA_LIB=$PWD/$v
  timeout -k 10 200 python bench.py --steps ${STEPS:-50} --warmup 3 --no-cpu-baseline $ARGS 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); r=d['roofline']; k=d.get('kernels_ms') or {}
print('%-36s step %7.1f us (long block %7.1f, without ep %7.1f)  warp %6.1f us  gram %6.1f us  intervals: prep %.1f warp %.1f A %.1f gram %.1f  sclk %s'%('$v', d['ms_per_step']*1e3, (d.get('ms_per_step_long') or 0)*1e3, (d['config'].get('no_ep_ms_per_step') or 0)*1e3, r['kernel_ms']*1e3, r['accumulate_kernel_ms']*1e3, k.get('prep_pose_texel',0)*1e3, k.get('warp',0)*1e3, k.get('post_warp_a',0)*1e3, k.get('gram',0)*1e3, (d.get('device') or {}).get('sysfs',{}).get('sclk_mhz')))" >> gpurun_out/variants.log
done; done
cat gpurun_out/variants.log
