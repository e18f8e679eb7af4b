mkdir -p gpurun_out; L=gpurun_out/r04_exp9.log; rm -f $L
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "state_parity or baseline_size or normal_equations or irls or randomised or other_configurations or panorama_border or resident_step" > gpurun_out/r04_tests2.log 2>&1; rc=$?
tail -3 gpurun_out/r04_tests2.log | tee -a $L
[ $rc -ne 0 ] && { grep -E "Error|assert|FAILED|error" gpurun_out/r04_tests2.log | head -30; exit $rc; }
run() { # label events pano_h K steps extra-args env...
  lbl=$1; n=$2; ph=$3; k=$4; st=$5; extra=$6; shift 6
  env "$@" timeout -k 10 400 python bench.py --steps $st --warmup 2 --no-cpu-baseline --events-per-gpu $n --pano-h $ph --knots $k $extra 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); r=d['roofline']; c=d['config']; s=c['setup']
print('%-34s N=%9d pano_h=%4d K=%3d: %7.3f G ev/s  step %9.1f us  warp %8.1f us  gram %8.1f us  frac %.3f | %s'%('$lbl', c['events_per_rank'], $ph, $k, d['value']/1e9, d['ms_per_step']*1e3, r['kernel_ms']*1e3, r['accumulate_kernel_ms']*1e3, r['frac'], 'tile' if s['tile_order'] else 'pixel'))" | tee -a $L
}
for sp in 0 1; do
run "1M segpose=$sp" 1000000 1024 21 200 "" EMBA_SEGPOSE=$sp
run "1.5M segpose=$sp" 1500000 1024 21 100 "" EMBA_SEGPOSE=$sp
run "10M 2048 K256 segpose=$sp" 10000000 2048 256 10 "" EMBA_SEGPOSE=$sp
run "10M 640x480 K97 segpose=$sp" 10000000 1024 97 10 "--sensor 640x480" EMBA_SEGPOSE=$sp
run "3M pixel K21 segpose=$sp" 3000000 1024 21 30 "" EMBA_SEGPOSE=$sp EMBA_ORDER=pixel
done
