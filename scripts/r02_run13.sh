set -e
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r13_tests.log 2>&1 || { tail -30 gpurun_out/r13_tests.log; exit 1; }
tail -2 gpurun_out/r13_tests.log
ORDERS="auto" bash scripts/scaling.sh
