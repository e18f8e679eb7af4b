mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sharded.py -m gpu -x -q 2>&1 | tail -3
for tg in 0 1 0 1; do EMBA_GRAM_TAGS=$tg timeout -k 10 200 python bench.py --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('1M uniform tags=$tg step %7.1f us  warp %6.1f us  gram %6.1f us'%(d['ms_per_step']*1e3, r['kernel_ms']*1e3, r['accumulate_kernel_ms']*1e3))"; done
for tg in 0 1; do EMBA_GRAM_TAGS=$tg timeout -k 10 200 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --data scene 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('1M scene   tags=$tg step %7.1f us  warp %6.1f us  gram %6.1f us'%(d['ms_per_step']*1e3, r['kernel_ms']*1e3, r['accumulate_kernel_ms']*1e3))"; done
for tg in 0 1; do EMBA_GRAM_TAGS=$tg timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --events-per-gpu 10000000 --knots 97 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('10M K=97   tags=$tg step %7.1f us  warp %6.1f us  gram %6.1f us'%(d['ms_per_step']*1e3, r['kernel_ms']*1e3, r['accumulate_kernel_ms']*1e3))"; done
