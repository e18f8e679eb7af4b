set -e
ARGS="--events-per-gpu 100000000 --knots 256 --pano-h 2048" STEPS=3 bash scripts/variants.sh
cp gpurun_out/variants.log gpurun_out/variants_100M.log
ARGS="--events-per-gpu 10000000 --knots 97" STEPS=10 bash scripts/variants.sh
cp gpurun_out/variants.log gpurun_out/variants_10M.log
ARGS="" STEPS=30 bash scripts/variants.sh
for ab in 0 256; do EMBA_ABLATE=$ab timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --events-per-gpu 100000000 --pano-h 2048 --knots 256 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('ablate %3d: warp %9.1f us  gram %8.1f us step %9.1f us'%($ab, r['kernel_ms']*1e3, r['accumulate_kernel_ms']*1e3, d['ms_per_step']*1e3))"; done
