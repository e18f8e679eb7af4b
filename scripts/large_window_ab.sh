# evidence (round 6): tile shape / origin grid options at 12.5 M - 100 M events on one box (profiles/r06_large_window_ab.txt)
mkdir -p gpurun_out
for cfg in "100M|--events-per-gpu 100000000 --knots 256 --pano-h 2048 --steps 4" "40M|--events-per-gpu 40000000 --knots 97 --pano-h 2048 --steps 5" "shard12M|--events-per-gpu 12500000 --knots 256 --pano-h 2048 --shard-of 8 --shard-rank 3 --steps 8"; do
  tag=${cfg%%|*}; args=${cfg#*|}
  for opts in "" "--opt tile_fine=0" "--opt tile_fine=0 --opt tile_shape=0" "--opt tile_fine=1"; do
    timeout -k 10 400 python bench.py --warmup 2 --no-cpu-baseline --no-with-ep --long-steps 0 $opts $args 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); r=d['roofline']; su=d['config']['setup']
print('%-8s %-36s warp %8.1f us  gram %8.1f us  step %8.1f us  entries %d chunks %d lead %.3f tile %s prep %.1f ms'%('$tag','$opts', r['kernel_ms_raw']*1e3, r['accumulate_kernel_ms']*1e3, d['ms_per_step']*1e3, su['entries'], su['chunks'], su['lead_in_frac'], su['tile'], su['prepare_ms']))"
  done
done
