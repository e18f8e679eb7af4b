# evidence (round 6): what producing the reference-order ep vector costs a step — tail blocks of the Gram launch (step_ep=1) vs launches of their own (step_ep=2) vs none (profiles/r06_ep_in_step.txt)
mkdir -p gpurun_out
for cfg in "city|--events-per-gpu 10000000 --knots 97 --sensor 640x480 --yaw-rate 0.1 --steps 10" "40M|--events-per-gpu 40000000 --knots 97 --pano-h 2048 --steps 5" "100M|--events-per-gpu 100000000 --knots 256 --pano-h 2048 --steps 4"; do
  tag=${cfg%%|*}; args=${cfg#*|}
  for opt in "step_ep=1" "step_ep=2"; do
  timeout -k 10 400 python bench.py --warmup 2 --no-cpu-baseline --long-steps 0 --opt $opt $args 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); c=d['config']; r=d['roofline']
print('$tag $opt  step %.1f us  no_ep %.1f us  with_ep(block) %.1f us  warp %.1f gram %.1f  ep_in_step %s'%(d['ms_per_step']*1e3, (c['no_ep_ms_per_step'] or 0)*1e3, (c['with_ep_ms_per_step'] or 0)*1e3, r['kernel_ms_raw']*1e3, r['accumulate_kernel_ms']*1e3, c['ep_in_step']))"
  done
done
