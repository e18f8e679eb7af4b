set -e
EMBA_ORDER=tile timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r12_tile_tests.log 2>&1 || { tail -30 gpurun_out/r12_tile_tests.log; exit 1; }
tail -2 gpurun_out/r12_tile_tests.log
ARGS="--events-per-gpu 100000000 --knots 256 --pano-h 2048" STEPS=3 bash scripts/variants.sh
cp gpurun_out/variants.log gpurun_out/variants_100M.log
ARGS="--events-per-gpu 10000000 --knots 97" STEPS=10 bash scripts/variants.sh
cp gpurun_out/variants.log gpurun_out/variants_10M.log
ARGS="--events-per-gpu 40000000 --knots 97 --pano-h 2048" STEPS=5 bash scripts/variants.sh
