# per-event pose in tile order: parity in forced tile order first, then the size sweep
set -e
mkdir -p gpurun_out
EMBA_ORDER=tile timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q > gpurun_out/r8_tile_tests.log 2>&1 || { tail -30 gpurun_out/r8_tile_tests.log; exit 1; }
tail -3 gpurun_out/r8_tile_tests.log
ORDERS="auto" bash scripts/scaling.sh
