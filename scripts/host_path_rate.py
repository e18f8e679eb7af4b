"""PCIe-inclusive rate of the drop-in (host-buffer) entry points: evaluateDataError(host Gx,Gy -> host ep, num_ev_map) +
formNormalEq + applyL2Reg with every block downloaded.  Reported in DESIGN.md; never bench.py's `value`."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emba_amd import LEGM
from emba_amd.synth import make_workload

w = make_workload()
m = LEGM(w.sensor_w, w.sensor_h, w.lut, w.C_th, w.pano_w, w.pano_h)
nem = np.zeros((w.pano_h, w.pano_w), dtype=np.int32)
t = time.perf_counter(); m.set_events(w.events); t_set = time.perf_counter() - t
ts = []
for it in range(12):
    t = time.perf_counter()
    ep = m.evaluateDataError(w.traj, w.Gx, w.Gy, None, True, nem)
    m.formNormalEq(ep, w.K, nem, w.thres_valid_pixel)
    ne = m.applyL2Reg(w.alpha)
    ts.append(time.perf_counter() - t)
med = float(np.median(ts[2:]))
print(f"host-buffer path: set_events {t_set*1e3:.1f} ms once; per iteration {med*1e3:.2f} ms = {w.events.size()/med/1e6:.1f} M events/s "
      f"(uploads 2x{w.Gx.nbytes/1e6:.0f} MB map, downloads ep {ep.nbytes/1e6:.1f} MB + count map {nem.nbytes/1e6:.0f} MB + blocks)")
