# evidence (round 6): pixel vs tile order on low-inlier / sparse large windows (profiles/r06_sparse_order_ab.txt); the full table: scripts/regime_sweep.py --set rule
for cfg in "10M_2048|--events-per-gpu 10000000 --knots 256 --pano-h 2048 --steps 10" "20M_2048|--events-per-gpu 20000000 --knots 256 --pano-h 2048 --steps 8" "10M_K97_fast|--events-per-gpu 10000000 --knots 97 --sensor 640x480 --steps 10"; do
  tag=${cfg%%|*}; args=${cfg#*|}
  for opts in "--opt order=1" "--opt order=2" ""; do
    timeout -k 10 400 python bench.py --warmup 2 --no-cpu-baseline --no-with-ep --long-steps 0 $opts $args 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); r=d['roofline']; su=d['config']['setup']
print('%-12s %-16s warp %8.1f us  gram %8.1f us  step %8.1f us  inl %.3f per_px %.1f lead %.3f %s tile %s'%('$tag','$opts', r['kernel_ms_raw']*1e3, r['accumulate_kernel_ms']*1e3, d['ms_per_step']*1e3, d['config']['inlier_frac'], su['events_per_pano_px'], su['lead_in_frac'], 'TILE' if su['tile_order'] else 'pixel', su['tile']))"
  done
done
