"""Time of the device Schur solve (emba_solve_normal_eq) next to the step it follows."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emba_amd import LEGM
from emba_amd.synth import make_workload
import os
CONFIGS = ((1_000_000, 1024, 21, 0.05), (1_000_000, 1024, 201, 0.005), (10_000_000, 1024, 97, 0.0104),
           (10_000_000, 1024, 201, 0.05))      # the last one: config 2's shape (shapes.launch: 10 s at dt = 0.05 s, K = 201) at the BASELINE event rate
if os.environ.get("SOLVE_CONFIGS"):
    CONFIGS = tuple(CONFIGS[int(i)] for i in os.environ["SOLVE_CONFIGS"].split(","))
for n, ph, K, dt in CONFIGS:
    w = make_workload(n_events=n, pano_h=ph, K=K, dt_knots=dt)
    m = LEGM(w.sensor_w, w.sensor_h, w.lut, w.C_th, w.pano_w, w.pano_h)
    for kv in os.environ.get("SOLVE_OPTS", "").split(","):      # e.g. SOLVE_OPTS=syrk_xcd=1 (emba_set_option)
        if "=" in kv:
            m.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    m.set_events(w.events); m.upload_map(w.Gx, w.Gy)
    ts, tv = [], []
    for it in range(6):
        t = time.perf_counter()
        m.eval_launch(w.traj); m.eval_finish(sync=False); m.form_active(w.thres_valid_pixel, sync=False); m.form_accumulate(); m.form_finish(w.alpha)
        n_inl, P = m.last_counts()
        ts.append(time.perf_counter() - t)
        t = time.perf_counter()
        x1, x2 = m.solveNormalEq(1e-3, fix_first_pose=True)
        tv.append(time.perf_counter() - t)
    print(f"N={n} K={K} P={P}: step {np.median(ts[1:])*1e3:.3f} ms, solve {np.median(tv[1:])*1e3:.3f} ms, |x1|={np.linalg.norm(x1):.3e} |x2|={np.linalg.norm(x2):.3e}")
    m.close()
