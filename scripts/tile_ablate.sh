# development: what the parts of the tiled warp kernel cost (diagnostics build, EMBA_ABLATE bit mask; results are wrong when non-zero)
#   /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -munsafe-fp-atomics -DEMBA_DIAG emba_amd/csrc/emba_hip.hip -o build_variants/diag.so
mkdir -p gpurun_out
export EMBA_LIB=$PWD/build_variants/diag.so
for cfg in "3M|--events-per-gpu 3000000 --steps 30" "city|--events-per-gpu 10000000 --knots 97 --sensor 640x480 --yaw-rate 0.1 --steps 12" "40M|--events-per-gpu 40000000 --knots 97 --pano-h 2048 --steps 5"; do
  tag=${cfg%%|*}; args=${cfg#*|}
  for ab in 0 2 32 8 4 16 1 46; do
    EMBA_ABLATE=$ab timeout -k 10 300 python bench.py --warmup 2 --no-cpu-baseline --no-with-ep --long-steps 0 --opt order=2 $args 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('%-6s ablate %-3s warp %8.1f us  gram %8.1f us  step %8.1f us'%('$tag','$ab', r['kernel_ms_raw']*1e3, r['accumulate_kernel_ms']*1e3, d['ms_per_step']*1e3))"
  done
done
