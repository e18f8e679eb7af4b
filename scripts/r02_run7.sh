mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/pytest.log
EMBA_ORDER=tile timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_tile.log 2>&1; echo "tile pytest rc=$?"; tail -2 gpurun_out/pytest_tile.log
ORDERS="auto" bash scripts/scaling.sh > /dev/null 2>&1; cat gpurun_out/scaling.log
