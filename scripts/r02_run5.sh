mkdir -p gpurun_out
timeout -k 10 300 python scripts/set_events_timing.py > gpurun_out/set_events_timing.txt 2>&1; cat gpurun_out/set_events_timing.txt
timeout -k 10 300 python bench.py --data scene --steps 50 --warmup 5 > gpurun_out/scene_bench.json 2> gpurun_out/scene_bench.err; tail -c 1500 gpurun_out/scene_bench.json
timeout -k 10 300 python bench.py --steps 50 --warmup 5 > gpurun_out/uniform_bench.json 2> gpurun_out/uniform_bench.err; tail -c 2000 gpurun_out/uniform_bench.json
