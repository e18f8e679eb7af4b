mkdir -p gpurun_out
EMBA_ORDER=tile timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_tile.log 2>&1; echo "tile-order rc=$?"; tail -25 gpurun_out/pytest_tile.log
ORDERS="tile" bash scripts/scaling.sh
