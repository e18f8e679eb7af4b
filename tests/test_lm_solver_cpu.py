"""The LM driver (emba_amd/solver.py, mirroring solver.cpp:11-368) run on the CPU oracle behind the same method names the
device LEGM exposes: checks the control flow itself; tests/test_gpu_parity.py runs the same loop on the HIP path."""
import numpy as np

from emba_amd import so3, synth
from emba_amd.legm import LinearTrajectory
from emba_amd.solver import BASettings, LMSettings, solve_time_window
from helpers import OracleModel


def perturbed(w, sigma=0.01, seed=5):
    rng = np.random.default_rng(seed)
    knots = w.traj.knots_xyzw.copy()
    for i in range(1, len(knots)):
        knots[i] = so3.mul(so3.exp(rng.normal(size=3) * sigma), knots[i])
    return LinearTrajectory(knots, w.traj.t0_ns, w.traj.dt_ns)


def knot_errors(a, b):
    return np.array([np.linalg.norm(so3.log(so3.mul(so3.inverse(p), q))) for p, q in zip(a.knots_xyzw, b.knots_xyzw)])


def test_event_simulator_is_consistent_with_the_measurement_model(oracle_mod):
    w = synth.make_scene_workload(n_steps=1000)
    assert w.events.size() > 20000 and (np.diff(w.events.t_ns) >= 0).all()
    m = OracleModel(oracle_mod, w)
    m.set_events(w.events)
    ep = m.evaluateDataError(w.traj, w.Gx, w.Gy)
    assert ep.size > 0.8 * w.events.size()
    assert np.sqrt((ep ** 2).mean()) < 0.5 * w.C_th          # events generated from the scene are explained by its gradient map


def test_lm_loop_on_oracle(oracle_mod):
    w = synth.make_scene_workload(n_steps=1000)
    init = perturbed(w)
    m = OracleModel(oracle_mod, w)
    r = solve_time_window(m, init, w.events, w.Gx, w.Gy, BASettings(alpha=0.0), LMSettings())
    costs = [c for (_, _, _, c, acc) in r.log if acc]
    assert len(costs) >= 3 and all(b < a for a, b in zip(costs, costs[1:]))
    assert r.cost_min == costs[-1] and r.cost_min < 0.5 * r.log[0][2]
    assert np.array_equal(r.traj.knots_xyzw[0], init.knots_xyzw[0])                       # first pose held (first_time_window)
    assert knot_errors(r.traj, w.traj).mean() < knot_errors(init, w.traj).mean()           # closer to the ground truth
    for (_, l10, _, _, _) in r.log:
        assert -300 <= l10 <= 3
    # lambda follows /10 on accept, x10 on reject (solver.cpp:330, 350)
    for (a, b) in zip(r.log, r.log[1:]):
        assert round(b[1] - a[1]) == (-1 if a[4] else 1)
    # IRLS variant runs and lowers its own (robust) cost
    r2 = solve_time_window(OracleModel(oracle_mod, w), init, w.events, w.Gx, w.Gy,
                           BASettings(use_IRLS=True, cost_type="huber", eta=0.1, alpha=1.0), LMSettings(max_num_iter=8))
    assert r2.cost_min < r2.log[0][2]


def test_runtime_log_files_follow_the_reference_format(oracle_mod, tmp_path):
    """solver.cpp:105-151, 170-178, 205-223, 271-291 + emba.cpp:223: the run-time records of a window, line for line in the reference's format."""
    import re
    from emba_amd.solver import RuntimeLog
    w = synth.make_scene_workload(n_steps=600)
    rl = RuntimeLog(str(tmp_path))
    r = solve_time_window(OracleModel(oracle_mod, w), perturbed(w), w.events, w.Gx, w.Gy, BASettings(alpha=1.0), LMSettings(max_num_iter=4), runtime_log=rl)
    d = tmp_path / "final_results"
    it = (d / "iterations.txt").read_text().splitlines()
    assert it[0] == "window #1" and set(it[1]) == {"-"}
    assert len(it) - 2 == r.iterations and all(re.fullmatch(r"iter #\d+:  log10\(lambda\) = \S+  cost_min\^2 = \S+  cost_new\^2 = \S+  cost_data = \S+  cost_reg = \S+", l) for l in it[2:])
    fe = (d / "runtime_formEqs.txt").read_text().splitlines()
    n_acc = sum(1 for e in r.log if e[4])
    assert len(fe) in (n_acc, n_acc + 1) and all(re.fullmatch(r"iter #\d+ count_formEqs = \d+ sec_total_formEqs = \S+ sec_average_formEqs = \S+", l) for l in fe)
    se = (d / "runtime_solveEqs.txt").read_text().splitlines()
    assert len(se) == r.iterations and se[-1].startswith(f"iter #{r.iterations - 1} count_solveEqs = {r.iterations} ")
    ob = (d / "runtime_objFuncs.txt").read_text().splitlines()
    assert len(ob) == r.iterations and all(re.fullmatch(r"iter #\d+ count_obj_func = \d+ sec_total_obj_func = \S+ sec_average_obj_func = \S+ Np = \S+", l) for l in ob)
    # a second window keeps counting (the reference's counters are function statics)
    solve_time_window(OracleModel(oracle_mod, w), perturbed(w), w.events, w.Gx, w.Gy, BASettings(alpha=1.0), LMSettings(max_num_iter=1), runtime_log=rl)
    assert (d / "iterations.txt").read_text().count("window #") == 2
    assert int((d / "runtime_solveEqs.txt").read_text().splitlines()[-1].split("count_solveEqs = ")[1].split()[0]) > r.iterations


def test_models_of_the_older_reject_contract_are_re_evaluated(oracle_mod):
    """a model whose rejectMap does not bring the equations back (no keeps_equations_on_reject) is evaluated and re-formed at the accepted
    point after a rejection — same decisions as the model that keeps them."""
    w = synth.make_scene_workload(n_steps=600)

    class Old(OracleModel):
        keeps_equations_on_reject = False
        n_eval = 0

        def rejectMap(self):
            self.trial = None      # (and nothing else: the state of the last evaluation stays the rejected trial's)

        def evaluateDataError(self, *a, **k):
            type(self).n_eval += 1
            return super().evaluateDataError(*a, **k)

    init = perturbed(w, sigma=0.03)
    ra = solve_time_window(OracleModel(oracle_mod, w, sparse=True), init, w.events, w.Gx, w.Gy, BASettings(alpha=1.0), LMSettings(max_num_iter=6))
    rb = solve_time_window(Old(oracle_mod, w, sparse=True), init, w.events, w.Gx, w.Gy, BASettings(alpha=1.0), LMSettings(max_num_iter=6))
    assert [e[4] for e in ra.log] == [e[4] for e in rb.log] and any(not e[4] for e in ra.log)
    assert np.allclose([e[3] for e in ra.log], [e[3] for e in rb.log], rtol=1e-9)
    assert Old.n_eval >= 1 + len(rb.log) + sum(1 for e in rb.log if not e[4])
