# development (round 6): the default library against build_variants/*.so of the same ABI on ONE box, interleaved (profiles/r06_early_gather_ab.txt, r06_shared_rcp_ab.txt)
#   LIBS="default sharedrcp" bash scripts/lib_ab.sh
mkdir -p gpurun_out
for rep in 1 2; do
for lib in ${LIBS:-default noearly}; do
  [ "$lib" = default ] && unset EMBA_LIB || export EMBA_LIB=$PWD/build_variants/$lib.so
  for cfg in "1M|--steps 200" "3M|--events-per-gpu 3000000 --steps 30" "5M_K97|--events-per-gpu 5000000 --knots 97 --steps 20" "city|--events-per-gpu 10000000 --knots 97 --sensor 640x480 --yaw-rate 0.1 --steps 12" "shard5M|--events-per-gpu 5000000 --knots 97 --sensor 640x480 --shard-of 8 --shard-rank 3 --yaw-rate 0.1 --steps 12" "40M|--events-per-gpu 40000000 --knots 97 --pano-h 2048 --steps 5"; do
    tag=${cfg%%|*}; args=${cfg#*|}
    timeout -k 10 300 python bench.py --warmup 2 --no-cpu-baseline --no-with-ep --long-steps 0 $args 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('%-8s %-10s warp %8.1f us  gram %8.1f us  step %8.1f us'%('$tag','$lib', r['kernel_ms_raw']*1e3, r['accumulate_kernel_ms']*1e3, d['ms_per_step']*1e3))"
  done
done
done
