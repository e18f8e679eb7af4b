"""Development / evidence: bench.py over the 1-10 M-event regimes under alternative options, one summary line per run
(profiles/r06_regime_sweep.txt).  Quick runs: no CPU baseline, no ep A/B blocks, no long block.

    python scripts/regime_sweep.py [--set quick|shapes|threshold|all] > gpurun_out/sweep.txt
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORK = {
    "1M": "",
    "1.25M": "--events-per-gpu 1250000",
    "1.5M": "--events-per-gpu 1500000",
    "1.75M": "--events-per-gpu 1750000",
    "2M": "--events-per-gpu 2000000",
    "2.5M": "--events-per-gpu 2500000",
    "3M": "--events-per-gpu 3000000",
    "5M_K97": "--events-per-gpu 5000000 --knots 97",
    "7M_K97": "--events-per-gpu 7000000 --knots 97",
    "10M_K97": "--events-per-gpu 10000000 --knots 97",
    "city": "--events-per-gpu 10000000 --knots 97 --sensor 640x480 --yaw-rate 0.1",
    "shard5M": "--events-per-gpu 5000000 --knots 97 --sensor 640x480 --shard-of 8 --shard-rank 3 --yaw-rate 0.1",
    "shard12M": "--events-per-gpu 12500000 --knots 256 --pano-h 2048 --shard-of 8 --shard-rank 3",
    "10M_2048": "--events-per-gpu 10000000 --knots 256 --pano-h 2048",
    "40M": "--events-per-gpu 40000000 --knots 97 --pano-h 2048",
}


def run(tag, work, opts, steps):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(steps), "--warmup", "3", "--no-cpu-baseline", "--no-with-ep", "--long-steps", "0"] + WORK[work].split()
    for o in opts:
        cmd += ["--opt", o]
    t0 = time.time()
    try:
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240, text=True)
        line = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ""
        d = json.loads(line)
    except Exception as e:   # noqa: BLE001
        print(f"{work:9s} {tag:28s} FAILED: {e!r}", flush=True)
        return None
    rf, cfg = d["roofline"], d["config"]
    su = cfg["setup"]
    tl = su.get("tile")
    print(f"{work:9s} {tag:28s} {d['value'] / 1e9:7.3f} G ev/s  step {d['ms_per_step'] * 1e3:8.1f} us  warp {rf['kernel_ms_raw'] * 1e3:7.1f} us (frac {rf['algorithmic_bytes_per_launch'] / (rf['kernel_ms_raw'] * 1e-3) / 8e12:5.3f})"
          f"  gram {rf['accumulate_kernel_ms'] * 1e3:7.1f} us  | {'tile' if su['tile_order'] else 'pixel'} entries {su['entries']} chunks {su['chunks']} lead {su['lead_in_frac']:.3f} "
          f"per_px {su['events_per_pano_px']:.1f} {('%dx%d@%dx%d r%d' % (tl['w'], tl['h'], tl['pitch_x'], tl['pitch_y'], tl['reserve'])) if tl else ''} inl {cfg['inlier_frac']:.3f} (predicted {su.get('inlier_frac_predicted', 0):.3f}) prep {su['prepare_ms']:.1f} ms  [{time.time() - t0:.0f} s]",
          flush=True)
    return d


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--set", default="quick")
    a = ap.parse_args()
    S = a.set
    if S in ("quick", "all"):
        for wk, steps in (("2M", 40), ("3M", 40), ("5M_K97", 30), ("city", 20), ("shard5M", 20), ("10M_K97", 20)):
            run("auto", wk, [], steps)
            run("tile forced", wk, ["order=2"], steps)
    if S in ("shapes", "all"):
        for wk, steps in (("2M", 40), ("3M", 40), ("5M_K97", 30), ("city", 20), ("shard5M", 20)):
            for sh in range(4):
                run(f"tile shape {sh}", wk, ["order=2", f"tile_shape={sh}"], steps)
            run("tile reserve 0", wk, ["order=2", "tile_reserve=0"], steps)
            run("tile fine", wk, ["order=2", "tile_fine=1"], steps)
            run("tile coarse", wk, ["order=2", "tile_fine=0"], steps)
    if S in ("threshold", "all"):
        for wk in ("1M", "1.25M", "1.5M", "1.75M", "2M", "2.5M", "3M"):
            run("pixel", wk, ["order=1"], 50)
            run("tile", wk, ["order=2"], 50)
    if S in ("chunks", "all"):
        for wk, steps in (("2M", 40), ("3M", 40), ("5M_K97", 30), ("city", 20)):
            for ch in (2016, 3024, 4032, 6048, 8064):
                run(f"tile chunk {ch}", wk, ["order=2", f"tile_chunk={ch}"], steps)
    if S in ("shares", "all"):
        for wk, steps in (("2M", 40), ("3M", 40), ("5M_K97", 30), ("city", 20), ("shard5M", 20), ("10M_K97", 20)):
            for sh in (0, 1, 2, 3, 4):
                run(f"tile shares {sh}", wk, ["order=2", f"tile_shares={sh}"], steps)
    if S in ("lib",):      # EMBA_LIB=build_variants/x.so python scripts/regime_sweep.py --set lib
        for wk, steps in (("2M", 40), ("3M", 40), ("5M_K97", 30), ("city", 20), ("shard5M", 20), ("10M_K97", 20), ("40M", 6)):
            run("tile " + os.path.basename(os.environ.get("EMBA_LIB", "default")), wk, ["order=2"], steps)
    if S in ("rule", "all"):      # does `auto` pick the faster order?  (every workload under auto, pixel and tile)
        WORK.update({"10M_fast": "--events-per-gpu 10000000 --knots 97 --sensor 640x480", "20M_2048": "--events-per-gpu 20000000 --knots 256 --pano-h 2048",
                     "5M_fast": "--events-per-gpu 5000000 --knots 97 --sensor 640x480", "4M_2048": "--events-per-gpu 4000000 --knots 256 --pano-h 2048"})
        for wk, steps in (("1.5M", 40), ("2M", 40), ("3M", 30), ("5M_K97", 20), ("city", 12), ("shard5M", 12), ("10M_K97", 12), ("10M_2048", 10), ("10M_fast", 10), ("20M_2048", 8), ("5M_fast", 12), ("4M_2048", 12), ("shard12M", 8)):
            for tag, o in (("auto", []), ("pixel", ["order=1"]), ("tile", ["order=2"])):
                run(tag, wk, o, steps)
    if S in ("large", "all"):
        for wk, steps in (("shard12M", 10), ("10M_2048", 10), ("40M", 6)):
            run("auto", wk, [], steps)


if __name__ == "__main__":
    main()
