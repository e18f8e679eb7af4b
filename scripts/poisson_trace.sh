# kernel trace of scripts/poisson_timing.py: per-kernel time of one emba_reconstruct_intensity at 2048x4096
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace_poisson
rm -rf $OUT && mkdir -p $OUT && cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/scripts/poisson_timing.py > $OUT/timing.log 2>&1 || { tail -5 $OUT/timing.log; exit 1; }
cat $OUT/timing.log | tail -4
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/*/*_kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# last 12 kernels that belong to the final reconstructIntensity call of the largest size (before any cleanup)
names = [r["Kernel_Name"].split("(")[0].replace("void ", "").replace("emba::", "") for r in rows]
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
idx = [i for i, n in enumerate(names) if "poisson" in n or "gemm" in n or "transpose" in n or "tridiag" in n or "thomas" in n or "sine" in n or "dst" in n or "div" in n.lower()]
last = idx[-14:]
t0 = int(rows[last[0]]["Start_Timestamp"])
for i in last:
    print("%-40s start %8.1f dur %7.1f grid %s" % (names[i][:40], (int(rows[i]["Start_Timestamp"]) - t0) / 1e3, dur[i], rows[i].get("Grid_Size_X", "")))
PY
