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


# the contract is PER VALUE ("float residuals/Jacobians within 1e-5 relative"); a norm-wise bound lets an element a million times
# below the largest be 100 % wrong.  Every comparison therefore also asserts |a_i - b_i| <= ELEM_RTOL |b_i| + ELEM_ATOL_REL max|b| element by element:
# 1e-5 relative per value, with an absolute floor of 1e-12 of the array's scale (sums with cancellation — A11, b1 — have entries that ARE rounding noise
# of their terms).  The worst element-wise errors seen per quantity are collected in WORST and printed at the end of the session (conftest.py).
ELEM_RTOL = 1e-5
ELEM_ATOL_REL = 1e-12
WORST = {}


def elementwise_err(a, b):
    """max over elements of |a_i - b_i| / (|b_i| + ELEM_ATOL_REL / ELEM_RTOL * max|b|): <= ELEM_RTOL iff every element meets the mixed bound."""
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    if a.size == 0:
        return 0.0
    s = float(np.max(np.abs(b)))
    if s == 0.0:
        return float(np.max(np.abs(a)))
    floor = ELEM_ATOL_REL / ELEM_RTOL * s
    worst = 0.0
    for lo in range(0, a.size, 1 << 24):          # (chunks: a 10 M x 12 array need not be copied three times)
        x = a.reshape(-1)[lo:lo + (1 << 24)]; y = b.reshape(-1)[lo:lo + (1 << 24)]
        worst = max(worst, float(np.max(np.abs(x - y) / (np.abs(y) + floor))))
    return worst


def assert_close_elementwise(a, b, what, rtol=ELEM_RTOL):
    e = elementwise_err(a, b)
    words = what.split()
    key = words[-1] if words[0].startswith("rank") else words[0]
    WORST[key] = max(WORST.get(key, 0.0), e)
    assert e <= rtol, f"{what}: an element is off by {e:.3e} relative (per-value bound {rtol:g} with an absolute floor of {ELEM_ATOL_REL:g} of the largest)"
    return e


def assert_close(a, b, what, tight=REL_TIGHT, elementwise=True):
    e = rel_err(a, b)
    assert e <= REL_CONTRACT, f"{what}: relative error {e:.3e} exceeds the 1e-5 contract"
    assert e <= tight, f"{what}: relative error {e:.3e} exceeds the engineering bound {tight:g}"
    if elementwise:
        assert_close_elementwise(a, b, what)
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


class OracleModel:
    """The CPU oracle behind the method names emba_amd.solver.solve_time_window drives, so the SAME LM loop can run on the
    oracle and on the device path and their iteration logs compared (test infrastructure only)."""

    keeps_equations_on_reject = True     # (rejectMap below brings the oracle's state back itself where it has to)

    def __init__(self, O, w, sparse=False, use_cg=False):
        """sparse: solve from the sparse A12 factors (sizes where the dense 3K x 2P matrix does not fit); use_cg: solveNormalEqCG."""
        self.O = O
        self.sparse, self.use_cg = sparse or use_cg, use_cg
        self.irls, self.a, self.thres = 0, 0.0, 5
        self.o = O.OracleLEGM(w.sensor_w, w.sensor_h, w.pano_w, w.pano_h, w.lut, w.C_th)
        self.H, self.W = w.pano_h, w.pano_w
        self.cur = self.trial = None

    def set_events(self, ev):
        self.ev = ev

    def evaluateDataError(self, traj, Gx, Gy, events=None, eval_deriv=True, num_ev_map=None):
        if Gx is not None:
            self.cur, self.trial = (np.array(Gx, dtype=np.float64), np.array(Gy, dtype=np.float64)), None
        self.used = self.trial if self.trial is not None else self.cur
        ev = self.ev
        self.ep, self.nem = self.o.evaluate_data_error(traj.knots_xyzw, traj.t0_ns, traj.dt_ns, self.used[0], self.used[1],
                                                       ev.x, ev.y, ev.polarity, ev.t_ns)
        self.K = traj.size()
        self._traj_eval = traj
        if num_ev_map is not None:
            num_ev_map[...] = self.nem
        return self.ep

    def dataCost(self, cost_type="quadratic", a=0.0):
        return self.O.data_cost(self.ep, {"quadratic": 0, "huber": 1, "cauchy": 2}[cost_type], a)

    def regCost(self, alpha):
        return self.O.reg_cost(self.used[0], self.used[1], alpha)

    def formNormalEq(self, ep, K, nem, thres):
        self._traj_form = self._traj_eval
        self.irls, self.a, self.thres = 0, 0.0, thres
        self.ne = self.o.form_normal_eq(self.ep, self.K, self.nem, thres, 0, 0.0, not self.sparse)

    def formNormalEqIRLS(self, ep, K, nem, thres, cost_type, a):
        self._traj_form = self._traj_eval
        self.irls, self.a, self.thres = {"quadratic": 0, "huber": 1, "cauchy": 2}[cost_type], a, thres
        self.ne = self.o.form_normal_eq(self.ep, self.K, self.nem, thres, self.irls, a, not self.sparse)

    def applyL2Reg(self, alpha):
        self.o.apply_l2(self.ne, alpha, self.used[0], self.used[1])

    def solveNormalEq(self, lam, fix_first_pose=False):
        if self.use_cg:
            return self.o.solve_cg_sparse(self.ne, self.ep, self.K, self.nem, self.thres, self.irls, self.a, lam, fix_first_pose)[:2]
        if self.sparse:
            return self.o.solve_sparse(self.ne, self.ep, self.K, self.nem, self.thres, self.irls, self.a, lam, fix_first_pose)
        return self.O.solve_normal_eq(self.ne, lam, fix_first_pose)

    def solveNormalEqCG(self, lam, fix_first_pose=False):
        return self.o.solve_cg_sparse(self.ne, self.ep, self.K, self.nem, self.thres, self.irls, self.a, lam, fix_first_pose)

    def updateMap(self, x2, damping):
        self.trial = self.O.update_map(self.ne["active"], x2, damping, self.cur[0], self.cur[1])

    def acceptMap(self):
        self.cur, self.trial = self.trial, None

    def rejectMap(self):
        self.trial = None
        if self.sparse:
            # The reference keeps A, b (dense A12 included) on the host across a rejected trial (solver.cpp:340-352).  The sparse solvers of
            # the oracle rebuild the A12 factors from the state of its LAST evaluateDataError — the rejected trial's — so that state is
            # brought back to the point the equations were formed at (test infrastructure: costs one CPU evaluation).
            self.evaluateDataError(self._traj_form, None, None)

    def downloadMap(self):
        return self.trial if self.trial is not None else self.cur
