"""Step time of the IRLS (huber / cauchy) variants of the resident step next to the quadratic one (BASELINE workload)."""
import sys, time
sys.path.insert(0, ".")
from emba_amd import LEGM
from emba_amd.synth import make_workload
w = make_workload()
m = LEGM(w.sensor_w, w.sensor_h, w.lut, w.C_th, w.pano_w, w.pano_h)
m.set_events(w.events); m.upload_map(w.Gx, w.Gy)
for cost, a in (("quadratic", 0.0), ("huber", 0.1), ("cauchy", 1.0)):
    for _ in range(3): m.step(w.traj, w.thres_valid_pixel, w.alpha, cost, a)
    m.sync(); t0 = time.perf_counter()
    for _ in range(30): m.step(w.traj, w.thres_valid_pixel, w.alpha, cost, a)
    m.sync(); dt = (time.perf_counter() - t0) / 30
    print(f"{cost:9s}: {dt * 1e6:7.1f} us/step = {w.events.size() / dt / 1e9:.2f} G events/s")
