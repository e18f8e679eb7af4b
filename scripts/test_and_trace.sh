# -m gpu tests, then rocprofv3 kernel-trace stats of the bench at 1 M and 100 M events (quick look at per-kernel times)
set -e
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r20_tests.log 2>&1 || { tail -30 gpurun_out/r20_tests.log; exit 1; }
tail -2 gpurun_out/r20_tests.log
cd /tmp && export TMPDIR=/tmp
for cfg in "1M:" "100M:--events-per-gpu 100000000 --knots 256 --pano-h 2048"; do
  tag=${cfg%%:*}; args=${cfg#*:}
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/trace_$tag -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline $args > $GRAFT_REPO_ROOT/gpurun_out/trace_$tag.log 2>&1
  python3 - <<PY
import csv,glob
f=glob.glob('$GRAFT_REPO_ROOT/gpurun_out/trace_$tag/*/*_kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:9]: print('$tag %-60s calls %4s avg %9.1f us'%(r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3))
PY
done
