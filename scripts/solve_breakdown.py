"""Per-kernel time of ONE device Schur solve per configuration, from a rocprofv3 kernel trace of scripts/solve_timing.py
(usage: python scripts/solve_breakdown.py gpurun_out/trace_solve)."""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].split("(")[0].replace("void ", "").replace("emba::", "").replace("emba_", "").replace("_kernel", "") for r in rows]
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
solves, i = [], 0
while i < len(rows):
    if names[i].startswith("csr_count"):
        j, agg = i, collections.OrderedDict()
        while j < len(rows):
            agg[names[j]] = agg.get(names[j], 0) + dur[j]
            if names[j].startswith("schur_x2"):
                break
            j += 1
        solves.append(((int(rows[j]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"])) / 1e3, agg))
        i = j
    i += 1
for k in range(len(solves)):
    if k % 6 == 3:
        wall, agg = solves[k]
        print("solve %6.0f us on the device: " % wall + ", ".join("%s %.0f" % (n, d) for n, d in agg.items()))
