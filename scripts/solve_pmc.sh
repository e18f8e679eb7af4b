# HBM traffic of the solve's kernels (FETCH_SIZE / WRITE_SIZE in separate passes) for one configuration of scripts/solve_timing.py
#   SOLVE_CONFIGS=3 bash scripts/solve_pmc.sh
OUT=$GRAFT_REPO_ROOT/gpurun_out/solve_pmc
rm -rf $OUT && mkdir -p $OUT && cd /tmp && export TMPDIR=/tmp
export SOLVE_CONFIGS=${SOLVE_CONFIGS:-3}
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 $GRAFT_REPO_ROOT/scripts/solve_timing.py > $OUT/fetch.log 2>&1 || { tail -5 $OUT/fetch.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 $GRAFT_REPO_ROOT/scripts/solve_timing.py > $OUT/write.log 2>&1 || { tail -5 $OUT/write.log; exit 1; }
python3 - <<'PY'
import csv, glob, collections, os
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/solve_pmc"
v = collections.defaultdict(lambda: collections.defaultdict(list))
for name in ("fetch", "write"):
    for r in csv.DictReader(open(glob.glob(f"{out}/{name}/*/*_counter_collection.csv")[0])):
        v[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/summary.txt", "w") as fo:
    for k, d in sorted(v.items(), key=lambda kv: -sum(kv[1]["FETCH_SIZE"])):
        f, w = d["FETCH_SIZE"], d["WRITE_SIZE"]
        line = "%-60s launches %4d  read %9.1f MB  written %9.1f MB per launch (2*FETCH_SIZE, WRITE_SIZE; KiB -> MB)" % (k, len(f), 2 * sum(f) / max(len(f), 1) * 1024 / 1e6, sum(w) / max(len(w), 1) * 1024 / 1e6)
        print(line); fo.write(line + "\n")
PY
rm -rf $OUT/fetch $OUT/write
