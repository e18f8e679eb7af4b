set -e
for rep in 1 2; do
ARGS="" STEPS=40 bash scripts/variants.sh; cp gpurun_out/variants.log gpurun_out/variants_1M_$rep.log
done
ARGS="--events-per-gpu 10000000 --knots 97 --sensor 640x480" STEPS=10 bash scripts/variants.sh; cp gpurun_out/variants.log gpurun_out/variants_10Mpx.log
ARGS="--events-per-gpu 10000000 --knots 97" STEPS=10 bash scripts/variants.sh; cp gpurun_out/variants.log gpurun_out/variants_10M.log
cat gpurun_out/variants_1M_1.log gpurun_out/variants_1M_2.log gpurun_out/variants_10Mpx.log gpurun_out/variants_10M.log
