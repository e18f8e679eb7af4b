mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1; rc=$?
tail -4 gpurun_out/gpu_tests.log
[ $rc -ne 0 ] && { grep -E "Error|assert|FAILED" gpurun_out/gpu_tests.log | head -30; exit $rc; }
timeout -k 10 600 python -m pytest tests -x -q -m "not gpu" > gpurun_out/cpu_tests.log 2>&1; tail -2 gpurun_out/cpu_tests.log
