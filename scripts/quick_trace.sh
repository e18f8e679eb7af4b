# kernel-trace stats of a short bench run -> gpurun_out/trace_$TAG/stats.txt (top kernels, average ns)
#   TAG=r03e ARGS="..." STEPS=200 bash scripts/quick_trace.sh
TAG=${TAG:-q}
ARGS=${ARGS:-}
STEPS=${STEPS:-200}
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace_$TAG
mkdir -p $OUT && cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps $STEPS --warmup 3 --no-cpu-baseline $ARGS > $OUT/bench.log 2>&1 || { tail -5 $OUT/bench.log; exit 1; }
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/trace/*/*_kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
with open("$OUT/stats.txt", "w") as fo:
    for r in rows[:12]:
        fo.write("%-60s calls %6s avg %10.1f ns  %5s %%\n" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]), r["Percentage"]))
print(open("$OUT/stats.txt").read())
PY
grep '^{' $OUT/bench.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], 'value', d['value'], 'warp_ms', d['roofline']['kernel_ms'], 'acc_ms', d['roofline']['accumulate_kernel_ms'])"
