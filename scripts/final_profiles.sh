# round-2 evidence: rocprofv3 kernel stats + counters of the three regimes (1 M: pixel order; 10 M / 100 M: tile order; scene events), sweep
mkdir -p gpurun_out
TAG=r02d_1M ARGS="" STEPS=50 bash scripts/profile.sh > gpurun_out/prof_1M.log 2>&1; echo "prof 1M rc=$?"
TAG=r02d_10M ARGS="--events-per-gpu 10000000 --knots 97" STEPS=10 bash scripts/profile.sh > gpurun_out/prof_10M.log 2>&1; echo "prof 10M rc=$?"
TAG=r02d_100M ARGS="--events-per-gpu 100000000 --knots 256 --pano-h 2048" STEPS=4 bash scripts/profile.sh > gpurun_out/prof_100M.log 2>&1; echo "prof 100M rc=$?"
TAG=r02d_scene ARGS="--data scene" STEPS=50 bash scripts/profile.sh > gpurun_out/prof_scene.log 2>&1; echo "prof scene rc=$?"
ORDERS="auto" bash scripts/scaling.sh > /dev/null 2>&1; cat gpurun_out/scaling.log
timeout -k 10 300 python bench.py > gpurun_out/bench_default.json 2>/dev/null; tail -c 900 gpurun_out/bench_default.json
