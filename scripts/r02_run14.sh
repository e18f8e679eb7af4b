set -e
EMBA_ORDER=tile timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r14_tile_tests.log 2>&1 || { tail -30 gpurun_out/r14_tile_tests.log; exit 1; }
tail -2 gpurun_out/r14_tile_tests.log
ORDERS="auto" bash scripts/scaling.sh
