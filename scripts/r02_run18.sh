set -e
ARGS="--events-per-gpu 100000000 --knots 256 --pano-h 2048" STEPS=3 bash scripts/variants.sh; cp gpurun_out/variants.log gpurun_out/variants_100M.log
ARGS="--events-per-gpu 10000000 --knots 97" STEPS=10 bash scripts/variants.sh; cp gpurun_out/variants.log gpurun_out/variants_10M.log
ARGS="" STEPS=40 bash scripts/variants.sh; cp gpurun_out/variants.log gpurun_out/variants_1M.log
cat gpurun_out/variants_100M.log gpurun_out/variants_10M.log gpurun_out/variants_1M.log
