"""emba_group_step with N rank contexts on ONE device (the only multi-rank timing a one-GPU box allows: the ranks' kernels run one after the
other on the card, the exchanges are the in-library kernel) — the ranks' forms as resident steps (emba_step_form_active: launch A with lists + zeroing, gather
inside the Gram launch, no clearing pass; group option group_step_fast = 1, default) against the sweeping forms of rounds 1-4 (= 0).
    python scripts/group_step_timing.py [ranks events_per_rank steps]"""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from emba_amd import _lib
from emba_amd.synth import make_workload
a = sys.argv[1:]
ranks, per, steps = (int(a[0]), int(a[1]), int(a[2])) if len(a) >= 3 else (8, 1_000_000, 40)
L = _lib.load()
w = make_workload(n_events=ranks * per)
cfg = _lib.EmbaCfg(w.sensor_w, w.sensor_h, w.pano_w, w.pano_h, w.lut.ctypes.data_as(C.POINTER(C.c_double)), w.C_th, 100, 10.0, 0, None)
def p(arr, t): return arr.ctypes.data_as(C.POINTER(t))
ev = w.events
x = np.ascontiguousarray(ev.x, np.uint16); y = np.ascontiguousarray(ev.y, np.uint16); pol = np.ascontiguousarray(ev.polarity, np.uint8); t = np.ascontiguousarray(ev.t_ns, np.int64)
k = np.ascontiguousarray(w.traj.knots_xyzw)
print(f"{w.describe()}: {ranks} rank contexts on device 0, {per} events per rank")
for n_r in sorted({1, 2, ranks}):
    sub = n_r * per
    dev = (C.c_int32 * n_r)(*([0] * n_r))
    g = C.c_void_p()
    assert L.emba_group_create(C.byref(cfg), dev, n_r, C.byref(g)) == 0
    assert L.emba_group_set_events(g, p(x, C.c_uint16), p(y, C.c_uint16), p(pol, C.c_uint8), p(t, C.c_int64), sub) == 0
    assert L.emba_group_upload_map(g, p(w.Gx, C.c_double), p(w.Gy, C.c_double)) == 0
    ni, P = C.c_size_t(0), C.c_size_t(0)
    res = {}
    for fast in (0, 1, 0, 1):
        if n_r > 1:
            assert L.emba_group_set_option(g, b"group_step_fast", fast) == 0
        for _ in range(10):
            assert L.emba_group_step(g, p(k, C.c_double), w.K, w.traj.t0_ns, w.traj.dt_ns, w.thres_valid_pixel, 0, 0.0, w.alpha, C.byref(ni), C.byref(P)) == 0, L.emba_group_last_error(g)
        t0 = time.perf_counter()
        for _ in range(steps):
            L.emba_group_step(g, p(k, C.c_double), w.K, w.traj.t0_ns, w.traj.dt_ns, w.thres_valid_pixel, 0, 0.0, w.alpha, C.byref(ni), C.byref(P))
        res.setdefault(fast, []).append((time.perf_counter() - t0) / steps * 1e3)
    if n_r == 1:
        print(f"  1 rank  (emba_step), {sub} events: {min(res[0] + res[1]):.3f} ms per step; inliers {ni.value}, P {P.value}")
    else:
        print(f"  {n_r} ranks, {sub} events: sweeping forms {min(res[0]):.3f} ms per step, resident-step forms {min(res[1]):.3f} ms per step "
              f"({min(res[1]) / n_r * 1e3:.1f} us per rank); inliers {ni.value}, P {P.value}")
    L.emba_group_destroy(g)
