# full validation of a build on the GPU box: -m gpu tests in both event orders, flip rate, then the profile set (scripts/final_profiles.sh)
set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r17_tests.log 2>&1 || { tail -30 gpurun_out/r17_tests.log; exit 1; }
tail -2 gpurun_out/r17_tests.log
EMBA_ORDER=tile timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r17_tile_tests.log 2>&1 || { tail -30 gpurun_out/r17_tile_tests.log; exit 1; }
tail -2 gpurun_out/r17_tile_tests.log
timeout -k 10 600 python scripts/flip_rate.py --out gpurun_out/r02_flip_rate.txt > gpurun_out/flip.log 2>&1 || { tail gpurun_out/flip.log; exit 1; }
tail -1 gpurun_out/r02_flip_rate.txt
bash scripts/final_profiles.sh
