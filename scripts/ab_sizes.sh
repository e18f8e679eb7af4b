# A/B of build_variants/*.so against the default build at the three HBM-regime sizes (same box, same run)
set -e
ARGS="--events-per-gpu 100000000 --knots 256 --pano-h 2048" STEPS=3 bash scripts/variants.sh; cp gpurun_out/variants.log gpurun_out/variants_100M.log
ARGS="--events-per-gpu 40000000 --knots 97 --pano-h 2048" STEPS=5 bash scripts/variants.sh; cp gpurun_out/variants.log gpurun_out/variants_40M.log
ARGS="--events-per-gpu 10000000 --knots 97" STEPS=10 bash scripts/variants.sh; cp gpurun_out/variants.log gpurun_out/variants_10M.log
cat gpurun_out/variants_100M.log gpurun_out/variants_40M.log gpurun_out/variants_10M.log
