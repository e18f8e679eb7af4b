mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1; rc=$?
tail -4 gpurun_out/gpu_tests.log
[ $rc -ne 0 ] && { grep -E "Error|assert|FAILED" gpurun_out/gpu_tests.log | head -30; exit $rc; }
timeout -k 10 600 python -m pytest tests -x -q -m "not gpu" > gpurun_out/cpu_tests.log 2>&1; tail -2 gpurun_out/cpu_tests.log
# the driver's multi-GPU command, rehearsed on the one-GPU box: six processes (the pool's process guard allows at most six of a job's processes on the card)
timeout -k 10 600 python3 bench.py --gpus 6 --one-device --steps 20 --warmup 5 > gpurun_out/r04_rehearsal_6ranks_one_device.json 2> gpurun_out/r04_rehearsal.err; echo "rehearsal rc=$?"; tail -c 1200 gpurun_out/r04_rehearsal_6ranks_one_device.json; tail -3 gpurun_out/r04_rehearsal.err
