mkdir -p gpurun_out
TAG=r04dbg ARGS="--events-per-gpu 1200000" STEPS=50 bash scripts/quick_trace.sh 2>&1 | tail -14
python scripts/step_timeline.py gpurun_out/trace_r04dbg/trace 2>&1 | tail -12
