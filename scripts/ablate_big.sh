# (needs a diagnostics build: the shipped library has no ablation hooks)
mkdir -p build_variants && [ -f build_variants/diag.so ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -munsafe-fp-atomics -DEMBA_DIAG emba_amd/csrc/emba_hip.hip -o build_variants/diag.so
export EMBA_LIB=$PWD/build_variants/diag.so
# diagnostics at 10 M events: Gram kernel under EMBA_ABLATE masks (timing only)
for a in ${ABLATES:-0 64 256 352}; do
  EMBA_ABLATE=$a timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --events-per-gpu 10000000 --pano-h 1024 --knots 97 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('ABLATE=$a  step %.1f us  warp %.1f us  gram %.1f us'%(d['ms_per_step']*1e3, r['kernel_ms']*1e3, r['accumulate_kernel_ms']*1e3))"
done
