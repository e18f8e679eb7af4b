"""Development aid (CPU only; uses the oracle as a pm calculator, like scripts/flip_rate.py): how many lead-in copies the tile order
needs under (a) the fixed-bin rule of rounds 2-5 (an entry belongs to the bw x bh bin its predicted pixel falls in; a chain that changes
bin costs a copy of the predecessor) and (b) the WINDOW rule of round 6 (a chain is cut greedily into the longest segments whose bounding
box still fits one LDS tile whose origin lies on the pitch grid).  Prints lead-in fraction and tiles in use per workload.

    python scripts/lead_in_sim.py                 # the regimes of profiles/r06_*
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from emba_amd.synth import make_workload  # noqa: E402
from emba_amd.sharded import shard_events  # noqa: E402
from oracle import oracle as O  # noqa: E402


def predicted_pixels(w, ev):
    o = O.OracleLEGM(w.sensor_w, w.sensor_h, w.pano_w, w.pano_h, w.lut, w.C_th)
    _, _, pm = o.count_map(w.traj.knots_xyzw, w.traj.t0_ns, w.traj.dt_ns, ev.x, ev.y, ev.t_ns, want_pm=True)
    return np.round(pm[:, 0]).astype(np.int32), np.round(pm[:, 1]).astype(np.int32)


def chains(ev, sensor_w):
    pix = ev.y.astype(np.int64) * sensor_w + ev.x
    order = np.argsort(pix, kind="stable")
    ps = pix[order]
    start = np.flatnonzero(np.r_[True, ps[1:] != ps[:-1]])
    length = np.diff(np.r_[start, ps.size])
    return order, start, length


def fixed_bins(px, py, order, start, length, bw, bh):
    b = (py[order] // bh).astype(np.int64) * 100000 + (px[order] // bw)
    brk = b[1:] != b[:-1]
    first = np.zeros(b.size, bool); first[start] = True
    n_break = int(np.count_nonzero(brk & ~first[1:]))
    return n_break, np.unique(b).size


def windows(px, py, order, start, length, tw, th, pitch_x, pitch_y, reserve):
    """greedy longest segments: bbox must fit [ox, ox + tw - 2 reserve) x [oy, ...) with ox = floor(xmin / pitch_x) * pitch_x"""
    X = px[order]; Y = py[order]
    wx, wy = tw - 2 * reserve, th - 2 * reserve
    nch = start.size
    xmin = X[start].copy(); xmax = xmin.copy(); ymin = Y[start].copy(); ymax = ymin.copy()
    n_break = 0
    tiles = set()
    tile_of = np.zeros(X.size, np.int64)
    seg_begin = start.copy()
    for k in range(1, int(length.max())):
        live = np.flatnonzero(length > k)
        i = start[live] + k
        nx0 = np.minimum(xmin[live], X[i]); nx1 = np.maximum(xmax[live], X[i])
        ny0 = np.minimum(ymin[live], Y[i]); ny1 = np.maximum(ymax[live], Y[i])
        ok = (nx1 - (nx0 // pitch_x) * pitch_x < wx) & (ny1 - (ny0 // pitch_y) * pitch_y < wy)
        cut = live[~ok]
        # close the segments that end here
        if cut.size:
            ids = (ymin[cut] // pitch_y).astype(np.int64) * 100000 + (xmin[cut] // pitch_x)
            tiles.update(np.unique(ids).tolist())
            n_break += cut.size
        xmin[live] = np.where(ok, nx0, X[i]); xmax[live] = np.where(ok, nx1, X[i])
        ymin[live] = np.where(ok, ny0, Y[i]); ymax[live] = np.where(ok, ny1, Y[i])
    ids = (ymin // pitch_y).astype(np.int64) * 100000 + (xmin // pitch_x)
    tiles.update(np.unique(ids).tolist())
    return n_break, len(tiles)


WORKLOADS = {
    "1.5M": dict(n=1_500_000), "2M": dict(n=2_000_000), "3M": dict(n=3_000_000), "5M_K97": dict(n=5_000_000, K=97),
    "city": dict(n=10_000_000, K=97, sensor=(640, 480), yaw=0.1),
    "shard5M": dict(n=40_000_000, K=97, sensor=(640, 480), yaw=0.1, shard=(3, 8)),
    "10M_K97": dict(n=10_000_000, K=97),
    "shard12M": dict(n=100_000_000, K=256, pano_h=2048, shard=(3, 8)),
}


def main():
    names = sys.argv[1:] or ["2M", "3M", "5M_K97", "city", "shard5M"]
    for name in names:
        c = WORKLOADS[name]
        t0 = time.time()
        sw, sh = c.get("sensor", (240, 180))
        w = make_workload(n_events=c["n"], pano_h=c.get("pano_h", 1024), K=c.get("K", 21), sensor=(sw, sh), focal=200.0 * sw / 240.0, yaw_rate=c.get("yaw", 0.5))
        ev = w.events
        if "shard" in c:
            ev, _ = shard_events(ev, w.sensor_w, *c["shard"])
        px, py = predicted_pixels(w, ev)
        order, start, length = chains(ev, w.sensor_w)
        n = ev.size()
        print(f"== {name}: {n} events, {start.size} chains, mean length {length.mean():.1f}  ({time.time() - t0:.0f} s)", flush=True)
        nb, nt = fixed_bins(px, py, order, start, length, 32, 8)
        print(f"   fixed bins 32x8 (rounds 2-5)                     lead-ins {nb / n:6.3f}   bins {nt}", flush=True)
        for (tw, th, pxp, pyp, r) in [(48, 24, 32, 8, 2), (48, 24, 32, 8, 0), (48, 24, 32, 8, 4), (48, 24, 16, 8, 2), (48, 24, 32, 4, 2), (48, 24, 16, 4, 2),
                                       (64, 18, 32, 8, 2), (72, 16, 32, 8, 2), (96, 12, 32, 4, 2), (36, 32, 16, 8, 2),
                                       (80, 32, 32, 8, 2), (80, 32, 64, 16, 2), (96, 24, 32, 8, 2), (128, 20, 64, 8, 2)]:
            nb, nt = windows(px, py, order, start, length, tw, th, pxp, pyp, r)
            print(f"   windows tile {tw:3d}x{th:2d} pitch {pxp:2d}x{pyp:2d} reserve {r}           lead-ins {nb / n:6.3f}   tiles {nt}   entries/tile {n * (1 + nb / n) / nt:8.0f}", flush=True)


if __name__ == "__main__":
    main()
