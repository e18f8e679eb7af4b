"""Where the host spends a step: wall time of each phase call (enqueue only until form_finish, which synchronizes)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emba_amd import LEGM
from emba_amd.synth import make_workload
w = make_workload()
m = LEGM(w.sensor_w, w.sensor_h, w.lut, w.C_th, w.pano_w, w.pano_h)
m.set_events(w.events); m.upload_map(w.Gx, w.Gy)
T = np.zeros(6)
for it in range(60):
    t = [time.perf_counter()]
    m.eval_launch(w.traj); t.append(time.perf_counter())
    m.eval_finish(sync=False); t.append(time.perf_counter())
    m.form_active(w.thres_valid_pixel, sync=False); t.append(time.perf_counter())
    m.form_accumulate(); t.append(time.perf_counter())
    m.form_finish(w.alpha); t.append(time.perf_counter())
    m.last_counts(); t.append(time.perf_counter())
    if it >= 10: T += np.diff(t)
T /= 50
print("us per call: eval_launch %.1f eval_finish %.1f form_active %.1f form_accumulate %.1f form_finish(sync) %.1f last_counts %.1f total %.1f" % (*(T * 1e6), T.sum() * 1e6))
