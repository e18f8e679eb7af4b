# kernel-trace stats of an arbitrary python script: SCRIPT=scripts/x.py TAG=name bash scripts/trace_cmd.sh
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace_${TAG:-x}
mkdir -p $OUT && cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/$SCRIPT > $OUT/run.log 2>&1 || { tail -3 $OUT/run.log; exit 1; }
python3 - <<PY
import csv,glob
for r in list(csv.reader(open(glob.glob('$OUT/*/*_kernel_stats.csv')[0])))[:${TOP:-14}]:
    print("%-70s %6s %14s %12s %6s"%(r[0][:70], r[1], r[2], r[3][:11], r[4][:6]))
PY
