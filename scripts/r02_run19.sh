set -e
ARGS="" STEPS=40 bash scripts/variants.sh; cp gpurun_out/variants.log gpurun_out/variants_1M.log
ARGS="--events-per-gpu 10000000 --knots 97 --sensor 640x480" STEPS=10 bash scripts/variants.sh; cp gpurun_out/variants.log gpurun_out/variants_10Mpx.log
ARGS="--data scene" STEPS=40 bash scripts/variants.sh; cp gpurun_out/variants.log gpurun_out/variants_scene.log
cat gpurun_out/variants_1M.log gpurun_out/variants_10Mpx.log gpurun_out/variants_scene.log
ORDERS=auto bash scripts/scaling.sh
