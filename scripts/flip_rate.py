#!/usr/bin/env python3
"""Measured flip rate of the index-level results: device path vs CPU oracle at 10 M - 100 M events.

For every configuration the device evaluates once (evaluateDataError), its count map (num_ev_map, model.cpp:227), its inlier
count and its per-event panorama coordinates pm are compared with the oracle's (emba_oracle_count_map: the same leaf
functions and pairing rule as the full oracle, O(sensor pixels) working memory):
  * pixels whose count differs and sum |count_dev - count_oracle| / 2  (one flipped measurement changes two pixels, or one
    if it flips between inlier and outlier)
  * events whose rounded pixel round(pm) differs (what model.cpp:209-211 indexes with)
  * how close pm is: share of bit-identical coordinates, largest difference in ulps
Writes a text report (default profiles/r02_flip_rate.txt).  Test infrastructure: never imported by the product.
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CONFIGS = {   # name: (events, sensor, pano_h, K)
    "1M-baseline": (1_000_000, (240, 180), 1024, 21),
    "10M-city": (10_000_000, (640, 480), 1024, 97),
    "40M-town": (40_000_000, (640, 480), 1024, 97),
    "100M-synthetic": (100_000_000, (240, 180), 2048, 256),
}


def ulp_diff(a, b):
    ai = a.view(np.int64); bi = b.view(np.int64)
    return np.abs(ai - bi)


def run(name, out):
    from emba_amd import LEGM
    from emba_amd.synth import make_workload
    from oracle import oracle as O
    n, sensor, pano_h, K = CONFIGS[name]
    t0 = time.perf_counter()
    w = make_workload(n_events=n, pano_h=pano_h, K=K, sensor=sensor, focal=200.0 * sensor[0] / 240.0)
    ev = w.events
    m = LEGM(w.sensor_w, w.sensor_h, w.lut, w.C_th, w.pano_w, w.pano_h, device=0)
    nem = np.zeros((w.pano_h, w.pano_w), dtype=np.int32)
    ep = m.evaluateDataError(w.traj, w.Gx, w.Gy, ev, True, nem)
    pm_dev = m.dump_state(fields=("pm",))["pm"]
    m.close()
    t1 = time.perf_counter()
    O.set_threads(min(O.max_threads(), 64))      # pm per event is independent of the thread count; the pairing pass is serial
    o = O.OracleLEGM(w.sensor_w, w.sensor_h, w.pano_w, w.pano_h, w.lut, w.C_th)
    n_inl, nem_o, pm_o = o.count_map(w.traj.knots_xyzw, w.traj.t0_ns, w.traj.dt_ns, ev.x, ev.y, ev.t_ns, want_pm=True)
    O.set_threads(1)
    t2 = time.perf_counter()
    n_used = (n // 100) * 100
    d = nem.astype(np.int64) - nem_o
    px_diff = int(np.count_nonzero(d))
    moved = float(np.abs(d).sum()) / 2.0
    pd, po = pm_dev[:n_used], pm_o[:n_used]
    rd = np.round(pd + 0.0); ro = np.round(po + 0.0)     # numpy rounds half to even; use the C rule instead:
    rd = np.where(pd >= 0, np.floor(pd + 0.5), -np.floor(-pd + 0.5)); ro = np.where(po >= 0, np.floor(po + 0.5), -np.floor(-po + 0.5))
    ev_flip = int(np.count_nonzero((rd != ro).any(axis=1)))
    ul = ulp_diff(np.ascontiguousarray(pd).ravel(), np.ascontiguousarray(po).ravel())
    same = float(np.count_nonzero(ul == 0)) / ul.size
    absd = float(np.abs(pd - po).max())
    max_ulp = absd / float(np.spacing(float(w.pano_w)))      # in ulps of the panorama width (pm.x wraps through 0 at the seam: raw ulps mean nothing there)
    frac = np.abs(pd - np.floor(pd) - 0.5)
    line = (f"{name:15s} events {n:>11,d} sensor {sensor[0]}x{sensor[1]} pano {pano_h}x{2*pano_h} K={K}: inliers dev {ep.size:,d} oracle {n_inl:,d} | "
            f"count-map pixels differing {px_diff}, measurements moved {moved:g} | events with a different rounded pixel {ev_flip} | "
            f"pm bit-identical {100*same:.4f} %, max |diff| {absd:.2e} px = {max_ulp:.1f} ulp(W), closest pm to a rounding boundary {frac.min():.3e} px | "
            f"workload + device {t1-t0:.1f} s, oracle {t2-t1:.1f} s")
    print(line, flush=True)
    out.write(line + "\n"); out.flush()
    return px_diff == 0 and ev_flip == 0 and ep.size == n_inl


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="1M-baseline,10M-city,40M-town,100M-synthetic")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r02_flip_rate.txt"))
    args = ap.parse_args()
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    ok = True
    with open(args.out, "w") as f:
        f.write("Flip rate of the index-level results, device path vs CPU oracle (scripts/flip_rate.py; one MI355X; fp contraction off on\n"
                "everything that feeds round(pm) and the outlier test; what is left is ocml vs glibc atan2/asin/sin/cos/atan)\n")
        for name in args.configs.split(","):
            ok = run(name, f) and ok
        f.write("all identical\n" if ok else "DIFFERENCES FOUND\n")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
