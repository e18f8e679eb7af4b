"""The caller of the hot path: EMBA::solveTimeWindow (reference src/emba/solver.cpp:11-368) — the Levenberg-Marquardt loop —
driving the device-resident path end to end: evaluateDataError / formNormalEq[IRLS] / applyL2Reg (hot path), solveNormalEq (f1),
updateMap + accept/reject (f2) all stay in HBM; only the 3K pose increments and the scalar costs cross to the host, where updateTraj
runs as in the reference.  Same constants and control flow as the reference: lambda 1e-3, x10 / /10, lambda in [1e-300, 1e3],
relative-cost tolerance counted over consecutive accepted steps."""
from dataclasses import dataclass, field

import numpy as np

from . import io as emba_io
from ._lib import ERR_NUMERIC as _ERR_NUMERIC


@dataclass
class LMSettings:               # include/emba/params.h:4-12, values of launch/shapes.launch:31-33
    max_num_iter: int = 50
    tol_fun: float = 1e-3
    num_times_tol_fun_sat: int = 2


@dataclass
class BASettings:               # include/emba/params.h:14-61, values of launch/shapes.launch:28-60
    use_IRLS: bool = False
    cost_type: str = "quadratic"
    eta: float = 0.1
    thres_valid_pixel: int = 5
    alpha: float = 5.0
    damping_factor: float = 1.0
    first_time_window: bool = True      # the first control pose is held fixed (solver.cpp:156-165, 227-230)
    use_CG: bool = False                # solveNormalEqCG instead of the Schur solve (solver.cpp:190-202; launch default false)


@dataclass
class LMResult:
    traj: object
    cost_min: float
    iterations: int
    converged: bool
    log: list = field(default_factory=list)      # (iter, log10(lambda), cost_min, cost_new, accepted)
    reason: str = ""            # which of the reference's stopping rules ended the loop: "tolerance" (solver.cpp:319-339), or the while condition of :63-64 —
                                # "lambda" (left [1e-300, 1e3]: no damping finds a better point), "cost" (below 1e-16), "max_iter"


class RuntimeLog:
    """The run-time records EMBA::solveTimeWindow appends under <result_dir>/final_results (solver.cpp:105-151 runtime_formEqs.txt,
    :205-223 runtime_solveEqs.txt, :271-291 runtime_objFuncs.txt, :196-202 CG_iterations.txt, :170-178 + emba.cpp:223 iterations.txt),
    in the reference's line formats.  Like its function-static counters, the totals run over all the windows of a process: keep ONE object per
    run and pass it to every solve_time_window.  The reference casts each duration to whole milliseconds before summing; the device path's
    phases are shorter than that, so the sums here are of the unrounded seconds (stated, not hidden: same fields, finer values)."""

    def __init__(self, result_dir):
        import os
        self.dir = os.path.join(result_dir, "final_results")
        os.makedirs(self.dir, exist_ok=True)
        self.n = dict(formEqs=0, solveEqs=0, obj_func=0)
        self.t = dict(formEqs=0.0, solveEqs=0.0, obj_func=0.0)
        self.window = 0
        open(os.path.join(self.dir, "iterations.txt"), "w").close()          # emba.cpp:223 opens it afresh

    def _app(self, name, line):
        import os
        with open(os.path.join(self.dir, name), "a") as f:
            f.write(line + "\n")

    def new_window(self):                                                    # solver.cpp:55-59
        self.window += 1
        self._app("iterations.txt", f"window #{self.window}")
        self._app("iterations.txt", "---------------------------------------------------------")

    def add(self, key, it, seconds, Np=None):
        self.n[key] += 1; self.t[key] += seconds
        n, tot = self.n[key], self.t[key]
        if key == "formEqs":
            self._app("runtime_formEqs.txt", f"iter #{it} count_formEqs = {n} sec_total_formEqs = {tot:.6g} sec_average_formEqs = {tot / n:.6g}")
        elif key == "solveEqs":
            self._app("runtime_solveEqs.txt", f"iter #{it} count_solveEqs = {n} sec_total_solveEqs = {tot:.9g} sec_average_solveEqs = {tot / n:.9g}")
        else:
            self._app("runtime_objFuncs.txt", f"iter #{it} count_obj_func = {n} sec_total_obj_func = {tot:.9g} sec_average_obj_func = {tot / n:.9g} Np = {Np}")

    def iteration(self, it, lam, cost_min, cost_new, cost_data, cost_reg):
        self._app("iterations.txt", f"iter #{it}:  log10(lambda) = {np.log10(lam):g}  cost_min^2 = {cost_min:g}  cost_new^2 = {cost_new:g}"
                                    f"  cost_data = {cost_data:g}  cost_reg = {cost_reg:g}")

    def cg(self, it, iters, err):
        self._app("CG_iterations.txt", f"iter #{it} iter_times = {iters} error = {err:g}")


class _Phases:
    """The three things the loop asks of the model, through the reference-shaped methods (evaluateDataError returns ep and the count
    map to the host, as LEGM::evaluateDataError does) or, with resident=True, through the phase-level calls that leave both in HBM."""

    def __init__(self, model, ba, resident):
        self.m, self.ba, self.resident = model, ba, resident
        self.cost_type = ba.cost_type if ba.use_IRLS else "quadratic"
        self.nem = None if resident else np.zeros((model.H, model.W), dtype=np.int32)

    def evaluate(self, traj, Gx=None, Gy=None):
        m = self.m
        if self.resident:
            if Gx is not None:
                m.upload_map(Gx, Gy)
            m.eval_launch(traj)
            if getattr(m, "async_phases", False) and hasattr(m, "costs"):
                m.eval_finish(sync=False)      # nothing is resolved on the host: the costs below are the synchronisation
            else:
                m.eval_finish()
        else:
            m.evaluateDataError(traj, Gx, Gy, None, True, self.nem)
        if hasattr(m, "costs"):      # both reductions, one synchronisation
            d, r = m.costs(self.cost_type, self.ba.eta, self.ba.alpha)
        else:
            d, r = m.dataCost(self.cost_type, self.ba.eta), m.regCost(self.ba.alpha)
        self.last_costs = (d, r)
        return d + r

    def form(self, K):
        m, ba = self.m, self.ba
        if self.resident:
            if getattr(m, "async_phases", False):
                m.form_active(ba.thres_valid_pixel, sync=False)      # P is resolved by the solve
            else:
                m.form_active(ba.thres_valid_pixel)
            m.form_accumulate(self.cost_type, ba.eta)
            m.form_finish(ba.alpha)
        else:
            if ba.use_IRLS:                                                          # :114-126 (ep stays on the device)
                m.formNormalEqIRLS(None, K, None, ba.thres_valid_pixel, ba.cost_type, ba.eta)
            else:
                m.formNormalEq(None, K, None, ba.thres_valid_pixel)
            m.applyL2Reg(ba.alpha)                                                   # :130


def solve_time_window(model, traj, events, Gx, Gy, ba=BASettings(), lm=LMSettings(), verbose=False, resident=False, runtime_log=None):
    """model: emba_amd.LEGM.  The refined map stays on the device (model.downloadMap()); returns LMResult.
    resident=True keeps residuals and count map in HBM too (only costs, counts and the 3K pose increments reach the host).
    runtime_log: a RuntimeLog — the reference's runtime_*.txt / iterations.txt records (each timed phase then ends in a host synchronisation,
    as it does in the reference's synchronous calls)."""
    import time
    rl = runtime_log
    if rl is not None:
        rl.new_window()

    def timed(key, it_, fn, Np=None):
        if rl is None:
            return fn()
        t0 = time.perf_counter()
        out = fn()
        if hasattr(model, "sync"):
            model.sync()
        rl.add(key, it_, time.perf_counter() - t0, Np() if callable(Np) else Np)
        return out
    lam, lam_max, lam_min = 1e-3, 1e3, 1e-300                                       # solver.cpp:15-17
    cost_min_old = cost_new = cost_min = 1e99
    it, count_tol, decreased = 0, 0, True
    ph = _Phases(model, ba, resident)
    model.set_events(events)
    if hasattr(model, "set_cost"):   # evaluations accumulate the IRLS-weighted sums directly (speed only)
        model.set_cost(ph.cost_type, ba.eta if ba.use_IRLS else 0.0)
    log = []
    while it <= lm.max_num_iter and cost_min > 1e-16 and lam_min <= lam <= lam_max:   # solver.cpp:63-64
        if decreased:
            if it == 0:                                                              # :69-91, uploads the initial map once
                cost_min = ph.evaluate(traj, Gx, Gy)
                cost_parts = ph.last_costs
            timed("formEqs", it, lambda: ph.form(traj.size()))                       # :93-131 (+ :105-151 runtime_formEqs.txt)
        if rl is not None:                                                           # :170-178
            rl.iteration(it, lam, cost_min, cost_new, *cost_parts)
        # x2 goes from the solver to updateMap and nowhere else (solver.cpp:193-239): a device model keeps it in HBM (x2 is None here)
        rkw = dict(resident_x2=True) if getattr(model, "supports_resident_x2", False) else {}
        try:
            if ba.use_CG:
                res = timed("solveEqs", it, lambda: model.solveNormalEqCG(lam, fix_first_pose=ba.first_time_window, **rkw))   # :196-202
                x1, x2 = res[:2]
                if rl is not None and len(res) >= 4:
                    rl.cg(it, res[2], res[3])
            else:
                x1, x2 = timed("solveEqs", it, lambda: model.solveNormalEq(lam, fix_first_pose=ba.first_time_window, **rkw))   # :190-194
        except Exception as e:   # noqa: BLE001
            # EMBA_ERR_NUMERIC: a 2x2 block A22_i + lambda*diag(A22_i) is not positive definite.  The reference's A22m_i.inverse()
            # (model.cpp:750) returns inf / nan there, x1 / x2 and the trial cost become NaN, `cost_new < cost_min` is false and the step is
            # rejected (solver.cpp:340-352): the same decision, without evaluating the NaN trial point.  (A vanishing pivot of S is not an
            # error any more: like Eigen's ldlt, model.cpp:789, the device solve returns a zero update for that component.)
            if getattr(e, "status", None) != _ERR_NUMERIC and "singular" not in str(e):
                raise
            it += 1
            log.append((it, float(np.log10(lam)), cost_min, float("inf"), False))
            decreased = False
            lam *= 10
            count_tol = 0
            continue
        traj_new = emba_io.incremental_update(traj, x1, ba.first_time_window)        # :226-234
        model.updateMap(x2, ba.damping_factor)                                       # :237-240 (trial map, on the device)
        n_act = (lambda: model.last_counts()[1]) if hasattr(model, "last_counts") else None
        cost_new = timed("obj_func", it + 1, lambda: ph.evaluate(traj_new), n_act)   # :251-268 (+ :271-291 runtime_objFuncs.txt, after iter += 1)
        cost_parts_new = ph.last_costs
        it += 1
        accepted = cost_new < cost_min
        log.append((it, float(np.log10(lam)), cost_min, cost_new, accepted))
        if verbose:
            print(f"iter #{it}: log10(lambda) = {np.log10(lam):.0f}  cost_min = {cost_min:.6e}  cost_new = {cost_new:.6e}  {'accept' if accepted else 'reject'}")
        if accepted:                                                                 # :299-339
            decreased = True
            traj = traj_new
            model.acceptMap()
            lam /= 10
            cost_min_old, cost_min = cost_min, cost_new
            cost_parts = cost_parts_new
            if abs(1 - cost_min / (cost_min_old + 1e-10)) < lm.tol_fun:
                count_tol += 1
                if count_tol >= lm.num_times_tol_fun_sat:
                    return LMResult(traj, cost_min, it, True, log, "tolerance")
        else:                                                                        # :340-352
            decreased = False
            # the reference reuses its host copies of A, b after a rejection (formNormalEq is skipped, solver.cpp:66-131).  On the device the
            # trial evaluation wrote a SECOND record set; rejectMap (emba_map_reject) makes the set the equations were formed from current
            # again — pack, active set and records untouched — so a rejection costs one evaluation, like in the reference.
            model.rejectMap()
            if not getattr(model, "keeps_equations_on_reject", False):
                # a model written to the older contract (its trial evaluation overwrites the state its equations were formed from): back to
                # the accepted point the long way — evaluate there again and re-form
                ph.evaluate(traj)
                ph.form(traj.size())
            lam *= 10
            count_tol = 0
    reason = "max_iter" if it > lm.max_num_iter else ("cost" if cost_min <= 1e-16 else "lambda")
    return LMResult(traj, cost_min, it, False, log, reason)
