"""Timeline of one BENCH step from a rocprofv3 kernel trace (usage: python scripts/step_timeline.py <trace dir>): start offsets, durations, gaps."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].split("(")[0].replace("void ", "").replace("emba::", "").replace("emba_", "").replace("_kernel", "") for r in rows]
idx = [i for i, n in enumerate(names) if n.startswith("prep_pose_texel")]
i0 = idx[len(idx) // 2]
t0 = int(rows[i0]["Start_Timestamp"])
prev_end = t0
for k in range(i0, i0 + 11):
    s, e = int(rows[k]["Start_Timestamp"]), int(rows[k]["End_Timestamp"])
    print("%-28s start %7.1f  dur %6.1f  gap before %5.1f" % (names[k][:28], (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3))
    prev_end = e
