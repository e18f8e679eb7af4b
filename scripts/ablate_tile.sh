# (needs a diagnostics build: the shipped library has no ablation hooks)
mkdir -p build_variants && [ -f build_variants/diag.so ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -munsafe-fp-atomics -DEMBA_DIAG emba_amd/csrc/emba_hip.hip -o build_variants/diag.so
export EMBA_LIB=$PWD/build_variants/diag.so
# anatomy of the tiled warp kernel by ablation (EMBA_ABLATE bits: 1 no marker, 2 no record stores, 4 no texel gather, 8 no per-pixel sums,
# 16 static camera = pose gathers hit two records); results are WRONG when non-zero, only the kernel time is read
mkdir -p gpurun_out; rm -f gpurun_out/ablate_tile.log
N=${N:-100000000}; PH=${PH:-2048}; K=${K:-256}
for ab in ${ABL:-0 8 2 4 16 10 14 30}; do
  EMBA_ORDER=${ORD:-tile} EMBA_ABLATE=$ab timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --events-per-gpu $N --pano-h $PH --knots $K 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('ablate %3d: warp %9.1f us  gram %8.1f us step %9.1f us'%($ab, r['kernel_ms']*1e3, r['accumulate_kernel_ms']*1e3, d['ms_per_step']*1e3))" >> gpurun_out/ablate_tile.log
done
cat gpurun_out/ablate_tile.log
