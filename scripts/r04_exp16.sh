# tile order: bin shape (LDS tile W x H incl. margin; same 1152-px budget): 48x24 m8 = bins of 32x8 (default), 72x16 m4 = 64x8, 56x20 m4 = 48x12, 40x28 m4 = 32x20, 64x18 m8 = 48x2
mkdir -p gpurun_out; L=gpurun_out/r04_exp16.log; rm -f $L
run() { lbl=$1; lib=$2; n=$3; k=$4; st=$5; extra=$6
  EMBA_LIB=$lib timeout -k 10 400 python bench.py --steps $st --warmup 2 --no-cpu-baseline --events-per-gpu $n --knots $k $extra 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); r=d['roofline']; c=d['config']; s=c['setup']
print('%-16s N=%9d K=%3d: step %9.1f us  warp %8.1f us  gram %8.1f us  frac %.3f | %s entries %d chunks %d'%('$lbl', c['events_per_rank'], $k, d['ms_per_step']*1e3, r['kernel_ms']*1e3, r['accumulate_kernel_ms']*1e3, r['frac'], 'tile' if s['tile_order'] else 'pixel', s['entries'], s['chunks']))" | tee -a $L
}
for v in default tile_72_16_4 tile_56_20_4 tile_40_28_4 tile_64_18_8; do
  lib=$PWD/build_variants/$v.so; [ $v = default ] && lib=$PWD/emba_amd/libemba_hip.so
  run $v $lib 3000000 21 40 ""
  run $v $lib 10000000 97 10 ""
  run $v $lib 40000000 97 5 "--pano-h 2048"
done
