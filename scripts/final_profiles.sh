# round-6 evidence (round 5: the same set under r05_*): rocprofv3 kernel stats + counters per regime, the solve / LM / host breakdowns.  Three parts (a gpurun call is at most 20 minutes):
#   PART=1 bash scripts/final_profiles.sh     1 M (BASELINE), SCALE's shard, the city shape on an inlier-rich stream, 10 M on 2048x4096, solve + LM + host timings
#   PART=2 ...                                3 M, 5 M, config 4's and 5's shards, 40 M, 100 M, scaling sweep
#   PART=3 ...                                flip rate, adapter timing, group step timing, default bench
mkdir -p gpurun_out gpurun_out/profiles_r06
P() { TAG=$1 ARGS="$2" STEPS=$3 bash scripts/profile.sh > gpurun_out/prof_$1.log 2>&1; echo "prof $1 rc=$?"; cp profiles/$1_* gpurun_out/profiles_r06/ 2>/dev/null; rm -rf gpurun_out/prof_$1/trace gpurun_out/prof_$1/fetch gpurun_out/prof_$1/write gpurun_out/prof_$1/atomic; }
if [ "${PART:-1}" = 1 ]; then
P r06_1M "" 50
P r06_shard1M "--shard-of 8 --shard-rank 3" 50
P r06_city "--events-per-gpu 10000000 --knots 97 --sensor 640x480 --yaw-rate 0.1" 10
P r06_10M_2048 "--events-per-gpu 10000000 --knots 256 --pano-h 2048" 10
TAG=r06 timeout -k 10 500 bash scripts/solve_trace.sh > gpurun_out/solve_trace_r06.log 2>&1; cp gpurun_out/trace_solve_r06/breakdown.txt gpurun_out/profiles_r06/r06_solve_breakdown_raw.txt; cat gpurun_out/trace_solve_r06/breakdown.txt
(timeout -k 10 300 python scripts/lm_timing.py; timeout -k 10 200 python scripts/lm_timing.py 1000000 1024 21 0.05 8) 2>&1 | grep -v amdgpu.ids > gpurun_out/profiles_r06/r06_lm_breakdown.txt; tail -12 gpurun_out/profiles_r06/r06_lm_breakdown.txt
(timeout -k 10 400 python scripts/resident_host_timing.py; timeout -k 10 200 python scripts/resident_host_timing.py 1000000 1024 21 0.05 8) 2>&1 | grep -v amdgpu.ids > gpurun_out/profiles_r06/r06_resident_host_timing.txt; cat gpurun_out/profiles_r06/r06_resident_host_timing.txt
fi
if [ "${PART:-1}" = 2 ]; then
P r06_2M "--events-per-gpu 2000000" 20
P r06_3M "--events-per-gpu 3000000" 20
P r06_5M "--events-per-gpu 5000000 --knots 97" 20
P r06_shard5M "--events-per-gpu 5000000 --knots 97 --sensor 640x480 --shard-of 8 --shard-rank 3 --yaw-rate 0.1" 10
P r06_shard12M "--events-per-gpu 12500000 --knots 256 --pano-h 2048 --shard-of 8 --shard-rank 3" 6
P r06_40M "--events-per-gpu 40000000 --knots 97 --pano-h 2048" 5
P r06_100M "--events-per-gpu 100000000 --knots 256 --pano-h 2048" 4
fi
if [ "${PART:-1}" = 3 ]; then
timeout -k 10 300 python bench.py > gpurun_out/profiles_r06/r06_bench_default.json 2>/dev/null; tail -c 600 gpurun_out/profiles_r06/r06_bench_default.json
timeout -k 10 300 python scripts/group_step_timing.py 2>&1 | grep -v amdgpu.ids > gpurun_out/profiles_r06/r06_group_step_8ranks_one_device.txt; cat gpurun_out/profiles_r06/r06_group_step_8ranks_one_device.txt
(timeout -k 10 400 python scripts/adapter_timing.py; timeout -k 10 200 python scripts/adapter_timing.py 1000000 1024 21 0.05 8) 2>&1 | grep -v amdgpu.ids > gpurun_out/profiles_r06/r06_adapter_timing.txt; cat gpurun_out/profiles_r06/r06_adapter_timing.txt
timeout -k 10 700 python scripts/flip_rate.py --out gpurun_out/profiles_r06/r06_flip_rate.txt > gpurun_out/flip.log 2>&1; echo "flip rate rc=$?"; tail -12 gpurun_out/profiles_r06/r06_flip_rate.txt
fi
du -sh gpurun_out
