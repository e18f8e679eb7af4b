mkdir -p gpurun_out
timeout -k 10 600 python scripts/flip_rate.py --out gpurun_out/r02_flip_rate.txt > gpurun_out/flip.log 2>&1; echo "flip rc=$?"
TAG=r02b_10M ARGS="--events-per-gpu 10000000 --knots 97" STEPS=10 bash scripts/profile.sh > gpurun_out/prof_10M.log 2>&1; echo "prof 10M rc=$?"
TAG=r02b_100M ARGS="--events-per-gpu 100000000 --knots 256 --pano-h 2048" STEPS=4 bash scripts/profile.sh > gpurun_out/prof_100M.log 2>&1; echo "prof 100M rc=$?"
ORDERS="auto" bash scripts/scaling.sh > /dev/null 2>&1; cat gpurun_out/scaling.log
EMBA_ORDER=tile timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
