for t in 1 0; do for a in "" "--data scene" "--events-per-gpu 10000000 --knots 97 --sensor 640x480"; do
EMBA_GRAM_TAGS=$t timeout -k 10 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline $a 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('tags=$t %-50s step %8.1f us warp %8.1f gram %8.1f'%('$a', d['ms_per_step']*1e3, r['kernel_ms']*1e3, r['accumulate_kernel_ms']*1e3))"
done; done
