# quick kernel-trace stats of the bench (development aid)
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace_${TAG:-x}
mkdir -p $OUT && cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench.log 2>&1 || { tail -3 $OUT/bench.log; exit 1; }
python3 - <<PY
import csv,glob
for r in list(csv.reader(open(glob.glob('$OUT/*/*_kernel_stats.csv')[0])))[:16]:
    print("%-62s %6s %12s %10s %6s"%(r[0][:62], r[1], r[2], r[3][:9], r[4][:6]))
PY
