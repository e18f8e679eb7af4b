# VALU cuts (polynomial sin/cos, Newton reciprocals in the Jacobian): parity in both orders, flip rate, size sweep
set -e
mkdir -p gpurun_out
EMBA_ORDER=tile timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r10_tile_tests.log 2>&1 || { tail -30 gpurun_out/r10_tile_tests.log; exit 1; }
tail -2 gpurun_out/r10_tile_tests.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r10_tests.log 2>&1 || { tail -30 gpurun_out/r10_tests.log; exit 1; }
tail -2 gpurun_out/r10_tests.log
timeout -k 10 600 python scripts/flip_rate.py --out gpurun_out/r02_flip_rate.txt > gpurun_out/flip.log 2>&1 || { tail gpurun_out/flip.log; exit 1; }
grep -i "flip\|differ\|identical" gpurun_out/r02_flip_rate.txt | head -30
ORDERS="auto" bash scripts/scaling.sh
