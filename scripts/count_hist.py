"""development: distribution of events per active panorama pixel for a solve_timing configuration (how long is the longest per-pixel record list?)"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emba_amd import LEGM
from emba_amd.synth import make_workload
n, ph, K, dt = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])) if len(sys.argv) > 4 else (10_000_000, 1024, 201, 0.05)
w = make_workload(n_events=n, pano_h=ph, K=K, dt_knots=dt)
m = LEGM(w.sensor_w, w.sensor_h, w.lut, w.C_th, w.pano_w, w.pano_h)
nem = np.zeros((w.pano_h, w.pano_w), dtype=np.int32)
m.evaluateDataError(w.traj, w.Gx, w.Gy, w.events, True, nem)
c = nem[nem >= w.thres_valid_pixel]
print("active", c.size, "mean", c.mean(), "max", c.max(), "p99", np.percentile(c, 99), "p99.9", np.percentile(c, 99.9), "count>256:", (c > 256).sum(), "count>1024:", (c > 1024).sum())
ys, xs = np.nonzero(nem == c.max())
print("argmax at", ys[:3], xs[:3])
