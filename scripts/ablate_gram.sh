# (needs a diagnostics build: the shipped library has no ablation hooks)
mkdir -p build_variants && [ -f build_variants/diag.so ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -munsafe-fp-atomics -DEMBA_DIAG emba_amd/csrc/emba_hip.hip -o build_variants/diag.so
export EMBA_LIB=$PWD/build_variants/diag.so
# development: Gram kernel anatomy at the BASELINE workload (EMBA_ABLATE bits: 32 no flush atomics, 64 no MFMA, 256 no activity lookups; results are WRONG)
for a in 0 32 64 256 288 352; do
  EMBA_ABLATE=$a timeout -k 10 200 python bench.py --steps ${STEPS:-200} --warmup 3 --no-cpu-baseline $ARGS 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('ablate %4s  step %9.1f us  warp %9.1f us  gram %8.1f us'%('$a', d['ms_per_step']*1e3, r['kernel_ms']*1e3, r['accumulate_kernel_ms']*1e3))"
done
