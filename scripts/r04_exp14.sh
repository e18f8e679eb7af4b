# pixel vs tile order around the threshold, after the gather moved beside the Gram stream (both orders use it at these sizes)
mkdir -p gpurun_out; L=gpurun_out/r04_exp14.log; rm -f $L
run() { lbl=$1; n=$2; st=$3; shift 3
  env "$@" timeout -k 10 400 python bench.py --steps $st --warmup 2 --no-cpu-baseline --events-per-gpu $n 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); r=d['roofline']; c=d['config']; s=c['setup']
print('%-22s N=%9d: %7.3f G ev/s  step %9.1f us  warp %8.1f us  gram %8.1f us  frac %.3f | %s'%('$lbl', c['events_per_rank'], d['value']/1e9, d['ms_per_step']*1e3, r['kernel_ms']*1e3, r['accumulate_kernel_ms']*1e3, r['frac'], 'tile' if s['tile_order'] else 'pixel'))" | tee -a $L
}
for n in 1500000 1750000 2000000 2500000 3000000; do
for o in pixel tile; do run "$o" $n 60 EMBA_ORDER=$o; done
done
