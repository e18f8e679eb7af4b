# round-4 evidence: rocprofv3 kernel stats + counters of every regime (1 M: pixel order; 3 M ... 100 M: tile order; scene events; the 8-GPU
# configurations' per-rank shards), the size sweep, the LM-loop breakdown, the flip rate
mkdir -p gpurun_out
P() { TAG=$1 ARGS="$2" STEPS=$3 bash scripts/profile.sh > gpurun_out/prof_$1.log 2>&1; echo "prof $1 rc=$?"; }
P r04_1M "" 50
P r04_3M "--events-per-gpu 3000000" 20
P r04_10M "--events-per-gpu 10000000 --knots 97" 10
P r04_10M_2048 "--events-per-gpu 10000000 --knots 256 --pano-h 2048" 10
P r04_40M "--events-per-gpu 40000000 --knots 97 --pano-h 2048" 5
P r04_100M "--events-per-gpu 100000000 --knots 256 --pano-h 2048" 4
P r04_scene "--data scene" 50
P r04_shard1M "--shard-of 8 --shard-rank 3" 50
P r04_shard5M "--events-per-gpu 5000000 --knots 97 --sensor 640x480 --shard-of 8 --shard-rank 3 --yaw-rate 0.1" 10
P r04_shard12M "--events-per-gpu 12500000 --knots 256 --pano-h 2048 --shard-of 8 --shard-rank 3" 6
ORDERS="auto" bash scripts/scaling.sh > /dev/null 2>&1; cp gpurun_out/scaling.log gpurun_out/r04_scaling_sweep.txt; cat gpurun_out/scaling.log
(python scripts/lm_timing.py; python scripts/lm_timing.py 1000000 1024 21 0.05 8) 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_lm_breakdown.txt; cat gpurun_out/r04_lm_breakdown.txt
timeout -k 10 300 python bench.py > gpurun_out/bench_default.json 2>/dev/null; tail -c 1500 gpurun_out/bench_default.json
timeout -k 10 900 python scripts/flip_rate.py --out gpurun_out/r04_flip_rate.txt > gpurun_out/flip.log 2>&1; echo "flip rate rc=$?"; tail -25 gpurun_out/r04_flip_rate.txt
# what comes back: the summaries (tracked under profiles/), not the raw rocprofv3 trees (gpurun merges at most 64 MiB of gpurun_out back)
mkdir -p gpurun_out/profiles_r04 && cp profiles/r04_* gpurun_out/profiles_r04/ 2>/dev/null
cp gpurun_out/r04_scaling_sweep.txt gpurun_out/r04_lm_breakdown.txt gpurun_out/r04_flip_rate.txt gpurun_out/bench_default.json gpurun_out/profiles_r04/ 2>/dev/null
for d in gpurun_out/prof_r04_*; do rm -rf $d/trace $d/fetch $d/write $d/atomic; done
du -sh gpurun_out
