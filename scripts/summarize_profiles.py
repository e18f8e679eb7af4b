"""Condense a gpurun_out/prof_<TAG> directory (scripts/profile.sh) into the tracked summaries under profiles/:
   profiles/<tag>_kernel_stats.csv     rocprofv3 --kernel-trace --stats (as written by rocprofv3)
   profiles/<tag>_pmc_summary.csv      per kernel: mean FETCH_SIZE / WRITE_SIZE (KiB) and corrected HBM bytes per launch
   profiles/<tag>_traffic.json         the dominant kernel's HBM bytes per launch, read by bench.py for roofline.traffic
gfx950 correction (MI355X_MICROARCH.md §HBM): FETCH_SIZE under-reports reads by exactly 2x; WRITE_SIZE is exact."""
import collections, csv, glob, json, os, shutil, sys

tag = sys.argv[1]
src = os.path.join("gpurun_out", f"prof_{tag}")
os.makedirs("profiles", exist_ok=True)
raw = glob.glob(f"{src}/trace/*/*_kernel_stats.csv")
if raw:
    shutil.copy(raw[0], f"profiles/{tag}_kernel_stats.csv")
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for name in ("fetch", "write", "atomic"):
    fs = glob.glob(f"{src}/{name}/*/*_counter_collection.csv")
    if not fs:
        continue
    for r in csv.DictReader(open(fs[0])):
        vals[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = []
for k, d in vals.items():
    f = sum(d["FETCH_SIZE"]) / max(len(d["FETCH_SIZE"]), 1); w = sum(d["WRITE_SIZE"]) / max(len(d["WRITE_SIZE"]), 1)
    at = sum(d["TCC_EA0_ATOMIC_sum"]) / max(len(d["TCC_EA0_ATOMIC_sum"]), 1)
    rows.append((k, len(d["FETCH_SIZE"]), f, w, (2 * f + w) * 1024, at))
rows.sort(key=lambda r: -r[4])
if not rows:      # the raw rocprofv3 trees were pruned (scripts/final_profiles.sh): work from the per-kernel summary written before that
    rows = [(r[0], int(r[1]), float(r[2]), float(r[3]), float(r[4]), float(r[5])) for r in list(csv.reader(open(f"profiles/{tag}_pmc_summary.csv")))[1:]]
with open(f"profiles/{tag}_pmc_summary.csv", "w") as fo:
    fo.write("kernel,launches,FETCH_SIZE_KiB_mean,WRITE_SIZE_KiB_mean,hbm_bytes_per_launch_corrected(2*FETCH+WRITE)*1024,TCC_EA0_ATOMIC_requests_mean\n")
    for r in rows:
        fo.write('"%s",%d,%.1f,%.1f,%.0f,%.0f\n' % r)
cand = [r for r in rows if "emba_warp_tiled_kernel" in r[0]] or [r for r in rows if "emba_warp_residual_kernel<false" in r[0]]
dom = cand[0]
bench = json.loads([l for l in open(f"{src}/bench_trace.log") if l.startswith("{")][-1])
wkey = bench["config"]["workload"]
par = bench["config"].get("parallelism", "")
if par.startswith("shard "):      # bench.py --shard-of N: one rank's part of the stream (bench.py looks the summary up under this key)
    t = par.split()
    wkey = f"{wkey} shard {t[1]} of {t[3]}"
json.dump({"kernel": "emba_warp_tiled_kernel" if "tiled" in dom[0] else "emba_warp_residual_kernel", "hbm_bytes_per_launch": dom[4], "FETCH_SIZE_KiB": dom[2], "WRITE_SIZE_KiB": dom[3],
           "atomic_requests_per_launch": dom[5], "correction": "gfx950: 2*FETCH_SIZE + WRITE_SIZE, KiB -> bytes", "workload": wkey, "tag": tag},
          open(f"profiles/{tag}_traffic.json", "w"), indent=1)
# the run itself could only look up an OLDER summary of this workload (or none): put the counters of THIS profile next to its kernel time
r = bench["roofline"]
r["traffic"] = dom[4]; r["counter_GBs"] = dom[4] / (r["kernel_ms"] * 1e-3) / 1e9; r["counter_frac"] = r["counter_GBs"] / r["peak"]
json.dump(bench, open(f"profiles/{tag}_bench_profiled.json", "w"))
print(open(f"profiles/{tag}_pmc_summary.csv").read())
