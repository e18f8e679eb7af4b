# round 4, experiment 4: cooperative active-write + Gram tail peel — targeted tests, A/B, trace, full tests
mkdir -p gpurun_out && rm -f gpurun_out/r04_exp4.log
L=gpurun_out/r04_exp4.log
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "resident_step or single_call_step or declared_cost or lm_loop or edge_cases or normal_equations or randomised or tile_order or other_configurations" > gpurun_out/r04_tests2.log 2>&1; rc=$?
tail -3 gpurun_out/r04_tests2.log | tee -a $L
[ $rc -ne 0 ] && { grep -E "Error|assert|FAILED|error" gpurun_out/r04_tests2.log | head -30; exit $rc; }
one() {  # label, env...
  lbl=$1; shift
  env "$@" timeout -k 10 200 python bench.py --steps ${STEPS:-400} --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('%-34s step %7.1f us  warp %6.1f us  gram %6.1f us'%('$lbl', d['ms_per_step']*1e3, r['kernel_ms']*1e3, r['accumulate_kernel_ms']*1e3))" | tee -a $L
}
one "default" X=1
one "EMBA_STEP_GATHER=1" EMBA_STEP_GATHER=1
one "EMBA_STEP_GATHER=0" EMBA_STEP_GATHER=0
one "EMBA_STEP_FAST=0" EMBA_STEP_FAST=0
one "GATHER=0 FAST=0 (round 3)" EMBA_STEP_GATHER=0 EMBA_STEP_FAST=0
one "default again" X=1
TAG=r04i STEPS=300 bash scripts/quick_trace.sh 2>&1 | tee -a $L
python scripts/step_timeline.py gpurun_out/trace_r04i/trace 2>&1 | tee -a $L
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r04_tests.log | tee -a $L
[ $rc -ne 0 ] && { grep -E "Error|assert|FAILED" gpurun_out/r04_tests.log | head -20; exit $rc; }
run() { # label events pano_h K steps extra-args env...
  lbl=$1; n=$2; ph=$3; k=$4; st=$5; extra=$6; shift 6
  env "$@" timeout -k 10 400 python bench.py --steps $st --warmup 2 --no-cpu-baseline --events-per-gpu $n --pano-h $ph --knots $k $extra 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); r=d['roofline']; c=d['config']; s=c['setup']
print('%-34s N=%9d pano_h=%4d K=%3d: %7.3f G ev/s  step %9.1f us  warp %8.1f us  gram %8.1f us  frac %.3f | %s entries %d chunks %d'%('$lbl', c['events_per_rank'], $ph, $k, d['value']/1e9, d['ms_per_step']*1e3, r['kernel_ms']*1e3, r['accumulate_kernel_ms']*1e3, r['frac'], 'tile' if s['tile_order'] else 'pixel', s['entries'], s['chunks']))" | tee -a $L
}
run "1.5M" 1500000 1024 21 100 "" X=1
run "1.5M gather0" 1500000 1024 21 100 "" EMBA_STEP_GATHER=0
run "shard 1M of 8M" 1000000 1024 21 100 "--shard-of 8 --shard-rank 3" X=1
run "scene" 1000000 1024 21 100 "--data scene" X=1
run "3M" 3000000 1024 21 50 "" X=1
run "3M gather0" 3000000 1024 21 50 "" EMBA_STEP_GATHER=0
run "5M K97" 5000000 1024 97 30 "" X=1
run "10M K97" 10000000 1024 97 20 "" X=1
run "10M K97 gather0" 10000000 1024 97 20 "" EMBA_STEP_GATHER=0
run "100M" 100000000 2048 256 5 "" X=1
exit 0
