# texel modes at 10 M events (pack = whole panorama per evaluation, rect = bounding box of the previous footprint)
for mode in pack rect; do
  for cfg in "1024 97" "2048 256"; do
    set -- $cfg
    EMBA_TEXEL=$mode timeout -k 10 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --events-per-gpu 10000000 --pano-h $1 --knots $2 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('$mode pano_h=$1 K=$2: step %.1f us  warp %.1f us  gram %.1f us'%(d['ms_per_step']*1e3, r['kernel_ms']*1e3, r['accumulate_kernel_ms']*1e3))"
  done
done
