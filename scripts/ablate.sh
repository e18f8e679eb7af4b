# (needs a diagnostics build: the shipped library has no ablation hooks)
mkdir -p build_variants && [ -f build_variants/diag.so ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -munsafe-fp-atomics -DEMBA_DIAG emba_amd/csrc/emba_hip.hip -o build_variants/diag.so
export EMBA_LIB=$PWD/build_variants/diag.so
# diagnostics: per-kernel times under EMBA_ABLATE bitmasks (results are wrong when non-zero; timing only)
mkdir -p gpurun_out; rm -f gpurun_out/ablate.log
for a in ${ABLATES:-0 1 2 4 8 15 32 64 96}; do
  echo "ABLATE=$a" >> gpurun_out/ablate.log
  EMBA_ABLATE=$a timeout -k 10 120 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('  step %.1f us  warp %.1f us  gram %.1f us'%(d['ms_per_step']*1e3, r['kernel_ms']*1e3, r['accumulate_kernel_ms']*1e3))" >> gpurun_out/ablate.log
done
cat gpurun_out/ablate.log
