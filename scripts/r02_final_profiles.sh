# round-2 evidence: tests, flip rate, sweeps, rocprofv3 kernel stats + counters of the three regimes (1 M: pixel order; 10 M / 100 M: tile order)
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/pytest.log
timeout -k 10 600 python scripts/flip_rate.py --out gpurun_out/r02_flip_rate.txt > gpurun_out/flip.log 2>&1; echo "flip rc=$?"
TAG=r02b_1M ARGS="" STEPS=50 bash scripts/profile.sh > gpurun_out/prof_1M.log 2>&1; echo "prof 1M rc=$?"
TAG=r02b_10M ARGS="--events-per-gpu 10000000 --knots 97" STEPS=10 bash scripts/profile.sh > gpurun_out/prof_10M.log 2>&1; echo "prof 10M rc=$?"
TAG=r02b_100M ARGS="--events-per-gpu 100000000 --knots 256 --pano-h 2048" STEPS=4 bash scripts/profile.sh > gpurun_out/prof_100M.log 2>&1; echo "prof 100M rc=$?"
TAG=r02b_scene ARGS="--data scene" STEPS=50 bash scripts/profile.sh > gpurun_out/prof_scene.log 2>&1; echo "prof scene rc=$?"
ORDERS="auto" bash scripts/scaling.sh > /dev/null 2>&1; cat gpurun_out/scaling.log
timeout -k 10 300 python scripts/set_events_timing.py 2>/dev/null | grep N= > gpurun_out/r02_set_events.txt; cat gpurun_out/r02_set_events.txt
