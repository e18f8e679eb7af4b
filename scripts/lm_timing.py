"""Wall time of the device LM loop (emba_amd.solver.solve_time_window = EMBA::solveTimeWindow, solver.cpp:63-353) per iteration, with the time
inside each model call, on the uniform synthetic workload of a configuration (events, pano_h, K, dt_knots) — default: config 2's shape.
    python scripts/lm_timing.py [n_events pano_h K dt_knots [max_iter]]"""
import sys, os, time, collections
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emba_amd import LEGM
from emba_amd.synth import make_workload, so3_exp_xyzw
from emba_amd.solver import BASettings, LMSettings, solve_time_window
a = sys.argv[1:]
n, ph, K, dt = (int(a[0]), int(a[1]), int(a[2]), float(a[3])) if len(a) >= 4 else (10_000_000, 1024, 201, 0.05)
max_iter = int(a[4]) if len(a) >= 5 else 6
w = make_workload(n_events=n, pano_h=ph, K=K, dt_knots=dt)
m = LEGM(w.sensor_w, w.sensor_h, w.lut, w.C_th, w.pano_w, w.pano_h)
for kv in filter(None, os.environ.get("EMBA_OPTS", "").split(",")):      # A/B: EMBA_OPTS=order=1,tile_reserve=4 python scripts/lm_timing.py
    name, _, val = kv.partition("=")
    m.set_option(name.strip(), int(val))
# a perturbed start (left-multiplied small rotations on every control pose but the first), so that the loop has something to do
rng = np.random.default_rng(3)
init = type(w.traj)(w.traj.knots_xyzw.copy(), w.traj.t0_ns, w.traj.dt_ns)
for i in range(1, K):
    e = so3_exp_xyzw(rng.normal(size=3) * 2e-3)
    ex, ey, ez, ew = e; bx, by, bz, bw = init.knots_xyzw[i]
    q = np.array([ew * bx + ex * bw + ey * bz - ez * by, ew * by + ey * bw + ez * bx - ex * bz, ew * bz + ez * bw + ex * by - ey * bx, ew * bw - ex * bx - ey * by - ez * bz])
    init.knots_xyzw[i] = q / np.linalg.norm(q)
T = collections.OrderedDict()
CALLS = {}
def timed(name):
    f = getattr(m, name)
    def g(*x, **k):
        t = time.perf_counter(); r = f(*x, **k); d = time.perf_counter() - t; T[name] = T.get(name, 0.0) + d; CALLS.setdefault(name, []).append(d); return r
    setattr(m, name, g)
for nm in ("set_events", "upload_map", "eval_launch", "eval_finish", "costs", "dataCost", "regCost", "form_active", "form_accumulate", "form_finish", "solveNormalEq", "updateMap",
           "acceptMap", "rejectMap"):
    if hasattr(m, nm): timed(nm)
# the host's own part of an iteration: updateTraj (solver.cpp:226-234) + what Python spends between the model calls
from emba_amd import io as emba_io
_iu = emba_io.incremental_update
def iu(*x, **k):
    t = time.perf_counter(); r = _iu(*x, **k); T["host: updateTraj"] = T.get("host: updateTraj", 0.0) + time.perf_counter() - t; return r
emba_io.incremental_update = iu
# two windows on the same model: the first one also pays for the context's buffers (hipMalloc of the record sets, sort workspaces ...: grow-only,
# kept for the next window); a bundle-adjustment run is a sequence of windows (emba.cpp), so the second is the steady state
for window in (1, 2):
    T.clear(); CALLS.clear()
    print(f"window {window}:")
    t0 = time.perf_counter()
    r = solve_time_window(m, init, w.events, w.Gx, w.Gy, BASettings(), LMSettings(max_num_iter=max_iter), resident=True)
    wall = time.perf_counter() - t0
    setup = T.get("set_events", 0) + T.get("upload_map", 0)
    print(f"N={n} K={K} pano {ph}x{2*ph}: {r.iterations} LM iterations, {sum(1 for e in r.log if e[4])} accepted; wall {wall*1e3:.1f} ms, of which set_events + first upload {setup*1e3:.1f} ms"
          f" -> {(wall-setup)/max(r.iterations,1)*1e3:.2f} ms per iteration")
    # the window's first evaluation also orders the events on the device (pixel / tile order: once per window, like set_events)
    el = CALLS.get("eval_launch", [0.0])
    once = max(0.0, el[0] - float(np.median(el[1:]))) if len(el) > 2 else 0.0
    print(f"   once per window: set_events + first upload {setup*1e3:.1f} ms, event ordering inside the first evaluation {once*1e3:.1f} ms"
          f" -> {(wall-setup-once)/max(r.iterations,1)*1e3:.2f} ms per iteration without them")
    print(f"   order: {m.setup_info()}; inliers outside their tile in the last evaluation, re-binnings of the window: {m.tile_drift()}")
    print("   ms inside the model calls: " + ", ".join(f"{k} {v*1e3:.2f}" for k, v in T.items()))
    inside = sum(v for k, v in T.items())
    print(f"   per iteration: " + ", ".join(f"{k} {v/max(r.iterations,1)*1e3:.3f}" for k, v in T.items() if k not in ("set_events", "upload_map")) +
          f", everything else (Python between the calls) {(wall - inside)/max(r.iterations,1)*1e3:.3f} ms")

m.close()
