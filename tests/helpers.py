"""Shared helpers for the parity tests."""
import numpy as np

# BASELINE north_star: "float residuals/Jacobians within 1e-5 relative".  fp64 on both sides agrees far better;
# the tests assert the contractual 1e-5 and ALSO a much tighter engineering bound so regressions show up early.
REL_CONTRACT = 1e-5
REL_TIGHT = 1e-9


def rel_err(a, b):
    """max |a-b| / max|b| (norm-wise relative error; 0 if both empty/zero)."""
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    if a.size == 0:
        return 0.0
    s = np.max(np.abs(b))
    d = np.max(np.abs(a - b))
    return 0.0 if d == 0 else d / (s if s > 0 else 1.0)


def assert_close(a, b, what, tight=REL_TIGHT):
    e = rel_err(a, b)
    assert e <= REL_CONTRACT, f"{what}: relative error {e:.3e} exceeds the 1e-5 contract"
    assert e <= tight, f"{what}: relative error {e:.3e} exceeds the engineering bound {tight:g}"
    return e


def small_workload(n_events=20000, pano_h=256, K=6, sensor=(64, 48), focal=60.0, seed=7, **kw):
    from emba_amd.synth import make_workload
    return make_workload(n_events=n_events, pano_h=pano_h, K=K, sensor=sensor, focal=focal, seed=seed, **kw)


def oracle_run(O, w, thres=None, irls=0, a=0.0, alpha=None, dense_A12=False, dump=False):
    """Full reference-order pass on the CPU oracle: evaluateDataError + formNormalEq[IRLS] + applyL2Reg."""
    o = O.OracleLEGM(w.sensor_w, w.sensor_h, w.pano_w, w.pano_h, w.lut, w.C_th)
    ev = w.events
    r = o.evaluate_data_error(w.traj.knots_xyzw, w.traj.t0_ns, w.traj.dt_ns, w.Gx, w.Gy, ev.x, ev.y, ev.polarity, ev.t_ns, dump=dump)
    ep, nem = r[0], r[1]
    ne = o.form_normal_eq(ep, w.K, nem, w.thres_valid_pixel if thres is None else thres, irls, a, dense_A12)
    al = w.alpha if alpha is None else alpha
    if al:
        o.apply_l2(ne, al, w.Gx, w.Gy)
    return dict(ep=ep, num_ev_map=nem, ne=ne, dump=r[2] if dump else None, oracle=o)
