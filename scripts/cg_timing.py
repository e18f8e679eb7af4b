import sys, os, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from emba_amd import LEGM
from emba_amd.synth import make_workload
for n, ph, K, dt in ((1_000_000, 1024, 21, 0.05), (10_000_000, 1024, 201, 0.05)):
    w = make_workload(n_events=n, pano_h=ph, K=K, dt_knots=dt)
    m = LEGM(w.sensor_w, w.sensor_h, w.lut, w.C_th, w.pano_w, w.pano_h)
    m.set_events(w.events); m.upload_map(w.Gx, w.Gy)
    m.eval_launch(w.traj); m.eval_finish(sync=False); m.form_active(w.thres_valid_pixel, sync=False); m.form_accumulate(); m.form_finish(w.alpha)
    for rep in range(3):
        t = time.perf_counter(); x1, x2 = m.solveNormalEq(1e-3, fix_first_pose=True); ts = time.perf_counter() - t
        t = time.perf_counter(); c1, c2, it, err = m.solveNormalEqCG(1e-3, fix_first_pose=True); tc = time.perf_counter() - t
    print(f"N={n} K={K}: Schur {ts*1e3:.2f} ms, CG {tc*1e3:.2f} ms for {it} iterations ({tc*1e3/max(it,1):.3f} ms each), error {err:.2e}, |x1 - x1_cg| / |x1| = {np.linalg.norm(x1-c1)/np.linalg.norm(x1):.2e}")
    m.close()
