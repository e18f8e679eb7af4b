"""CPU tests that PIN the oracle (test infrastructure) before it is trusted as the checker.

 * a3 (SO(3) linear spline): bit-for-bit against tests/golden/so3_spline_n2.npz, generated from the reference's own
   unmodified basalt/Sophus/Eigen headers (tests/golden/make_so3_spline_golden.py), and — where oracle/_ref is present —
   live against that reference build, including exp/log/left-Jacobian branches.
 * everything else is "parity unpinned" (no reference fixtures exist): checked by numeric differentiation in
   the style of basalt's test_spline.cpp:95-131 / test_utils.h:22-60, by the structural identities the reference guarantees
   (J0 + J1 = I; dp_norm > 10 rejection; quirk Q1 tail drop) and by independent implementations (scipy Sobel).
"""
import os

import numpy as np
import pytest

from helpers import small_workload, oracle_run

GOLD = os.path.join(os.path.dirname(__file__), "golden", "so3_spline_n2.npz")


def test_spline_matches_reference_golden_bitwise(oracle_mod):
    O = oracle_mod
    g = np.load(GOLD)
    t0, dt = int(g["t0_ns"]), int(g["dt_ns"])
    tags = set()
    for i in range(g["t"].size):
        r = O.spline_eval(g["knots"][i], t0, dt, int(g["t"][i]))
        assert r is not None
        q, R, s, J = r
        assert s == int(g["s"][i])                      # integer control-pose index: exact
        assert np.array_equal(q, g["q"][i]), (i, g["tag"][i])
        assert np.array_equal(R, g["R"][i]), (i, g["tag"][i])
        assert np.array_equal(J, g["J"][i]), (i, g["tag"][i])
        tags.add(str(g["tag"][i]))
    assert tags == {"general", "tiny", "identical", "on_knot", "near_pi"}


def test_spline_structural_identity_J0_plus_J1_is_I(oracle_mod):
    g = np.load(GOLD)
    for i in range(0, g["t"].size, 7):
        _, _, _, J = oracle_mod.spline_eval(g["knots"][i], int(g["t0_ns"]), int(g["dt_ns"]), int(g["t"][i]))
        assert np.allclose(J[:, :3] + J[:, 3:], np.eye(3), atol=1e-15)


def test_spline_rejects_times_outside_knots(oracle_mod):
    g = np.load(GOLD)
    k, t0, dt = g["knots"][0], int(g["t0_ns"]), int(g["dt_ns"])
    K = k.shape[0]
    assert oracle_mod.spline_eval(k, t0, dt, t0 - 1) is None
    assert oracle_mod.spline_eval(k, t0, dt, t0 + dt * (K - 1)) is None      # s + 2 > K
    assert oracle_mod.spline_eval(k, t0, dt, t0 + dt * (K - 1) - 1) is not None


@pytest.mark.skipif(not __import__("oracle.oracle", fromlist=["x"]).ref_available(), reason="oracle/_ref not built/shipped")
def test_primitives_match_reference_build_bitwise(oracle_mod):
    O = oracle_mod
    rng = np.random.default_rng(5)
    for scale in (1e-12, 1e-7, 1e-4, 0.05, 0.7, 2.0, 3.0):
        for _ in range(200):
            w = rng.normal(size=3)
            w = w / np.linalg.norm(w) * scale * rng.uniform(0.5, 1.0)
            assert np.array_equal(O.so3_exp(w), O.so3_exp(w, use_ref=True))
            q = O.so3_exp(w, use_ref=True)
            assert np.array_equal(O.so3_log(q), O.so3_log(q, use_ref=True))
            a, b = O.left_jacobian(w), O.left_jacobian(w, use_ref=True)
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    # the Taylor switch of Jl/Jl^-1 sits at |phi|^2 = 1e-10 (sophus_utils.hpp:351,392)
    for n in (0.99e-5, 1.01e-5):
        w = np.array([n, 0, 0.0])
        a, b = O.left_jacobian(w), O.left_jacobian(w, use_ref=True)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def _num_jac_left_knot(O, knots, t0, dt, t, which, eps=1e-6):
    """basalt testEvaluateSo3 (test_spline.cpp:95-131): left perturbation exp(x)*knot, residual log(res1*res^-1)."""
    q0, R0, s, _ = O.spline_eval(knots, t0, dt, t)
    J = np.zeros((3, 3))
    for j in range(3):
        vals = []
        for sgn in (+1, -1):
            d = np.zeros(3); d[j] = sgn * eps
            kn = knots.copy()
            e = O.so3_exp(d)
            a = kn[s + which]
            # exp(d) * knot
            ex, ey, ez, ew = e; bx, by, bz, bw = a
            kn[s + which] = np.array([ew * bx + ex * bw + ey * bz - ez * by, ew * by + ey * bw + ez * bx - ex * bz,
                                      ew * bz + ez * bw + ex * by - ey * bx, ew * bw - ex * bx - ey * by - ez * bz])
            _, R1, _, _ = O.spline_eval(kn, t0, dt, t)
            dR = R1 @ R0.T
            vals.append(np.array([dR[2, 1] - dR[1, 2], dR[0, 2] - dR[2, 0], dR[1, 0] - dR[0, 1]]) / 2)  # vee(log) to 1st order
        J[:, j] = (vals[0] - vals[1]) / (2 * eps)
    return J


def test_spline_jacobian_vs_numeric_differentiation_order2(oracle_mod):
    """Upstream basalt only instantiates this check for N=4,5,6 (test_spline.cpp:493-523); N=2 is what EMBA uses."""
    g = np.load(GOLD)
    t0, dt = int(g["t0_ns"]), int(g["dt_ns"])
    idx = [i for i in range(g["t"].size) if str(g["tag"][i]) == "general"][:25]
    for i in idx:
        knots, t = g["knots"][i].copy(), int(g["t"][i])
        _, _, s, J = oracle_mod.spline_eval(knots, t0, dt, t)
        for which in (0, 1):
            Jn = _num_jac_left_knot(oracle_mod, knots, t0, dt, t, which)
            assert np.allclose(J[:, 3 * which:3 * which + 3], Jn, atol=1e-3, rtol=0), (i, which)  # test_utils.h tolerance


def test_projection_jacobian_vs_numeric_differentiation(oracle_mod):
    rng = np.random.default_rng(11)
    for W, H in ((2048, 1024), (512, 256)):
        for _ in range(50):
            rb = rng.normal(size=3); rb[2] = abs(rb[2]) + 0.2
            pm, J = oracle_mod.project(W, H, rb)
            Jn = np.zeros((2, 3))
            for j in range(3):
                d = np.zeros(3); d[j] = 1e-6
                Jn[:, j] = (oracle_mod.project(W, H, rb + d)[0] - oracle_mod.project(W, H, rb - d)[0]) / 2e-6
            assert np.allclose(J, Jn, rtol=1e-5, atol=1e-4)
            assert 0 < pm[0] <= W and 0 <= pm[1] <= H      # equirectangular_camera.h:44 range


def test_projection_known_answers(oracle_mod):
    W, H = 2048, 1024
    pm, J = oracle_mod.project(W, H, np.array([0.0, 0.0, 1.0]))
    assert np.array_equal(pm, [1024.0, 512.0])                      # optical axis -> panorama centre
    fx = (W / 360.0) * 180.0 / np.pi
    assert fx == pytest.approx(325.94932345220167, rel=1e-15)       # SURVEY §8c probe value at 2048x1024
    pm, _ = oracle_mod.project(W, H, np.array([1.0, 0.0, 1.0]))     # 45 degrees of yaw
    assert pm[0] == pytest.approx(1024 + fx * np.pi / 4, rel=1e-15) and pm[1] == 512.0
    assert J[0, 0] == pytest.approx(fx, rel=1e-15) and J[0, 1] == 0.0


def test_warp_jacobian_vs_numeric_differentiation(oracle_mod):
    """d pm / d(left perturbation of rot) = J23 (event_pano_warper.cpp:62-65)."""
    O = oracle_mod
    w = small_workload(n_events=100)
    o = O.OracleLEGM(w.sensor_w, w.sensor_h, w.pano_w, w.pano_h, w.lut, w.C_th)
    rng = np.random.default_rng(3)
    for _ in range(30):
        q = O.so3_exp(rng.normal(size=3) * 0.3)
        x, y = int(rng.integers(0, w.sensor_w)), int(rng.integers(0, w.sensor_h))
        pm, J = o.warp(x, y, q)
        Jn = np.zeros((2, 3))
        for j in range(3):
            vals = []
            for sgn in (1, -1):
                d = np.zeros(3); d[j] = sgn * 1e-6
                e = O.so3_exp(d)
                ex, ey, ez, ew = e; bx, by, bz, bw = q
                q2 = np.array([ew * bx + ex * bw + ey * bz - ez * by, ew * by + ey * bw + ez * bx - ex * bz,
                               ew * bz + ez * bw + ex * by - ey * bx, ew * bw - ex * bx - ey * by - ez * bz])
                vals.append(o.warp(x, y, q2)[0])
            Jn[:, j] = (vals[0] - vals[1]) / 2e-6
        assert np.allclose(J, Jn, rtol=1e-5, atol=1e-4)


def test_batch_midpoint_ros_time_semantics(oracle_mod):
    O = oracle_mod
    assert O.batch_mid_ns(100, 100) == 100
    assert O.batch_mid_ns(1_000_000_000, 1_000_000_100) == 1_000_000_050
    assert O.batch_mid_ns(0, 3) == 2                      # 1.5 ns rounds half away from zero (Duration::fromSec)
    assert O.batch_mid_ns(0, 1) == 1                      # 0.5 ns -> 1
    assert O.batch_mid_ns(999_999_999, 3_000_000_001) == 2_000_000_000
    rng = np.random.default_rng(0)
    for _ in range(1000):
        a = int(rng.integers(0, 2**40)); d = int(rng.integers(0, 2**33))
        m = O.batch_mid_ns(a, a + d)
        assert abs(m - (a + d / 2)) <= 1.0                # within 1 ns of the exact midpoint


def test_sobel_hessian_against_independent_implementation(oracle_mod):
    ndi = pytest.importorskip("scipy.ndimage")
    rng = np.random.default_rng(2)
    for shape in ((7, 14), (32, 64), (1, 2), (2, 4)):
        Gx, Gy = rng.normal(size=shape), rng.normal(size=shape)
        Gxx, Gxy, Gyy = oracle_mod.hessian(Gx, Gy)
        # scipy 'mirror' == OpenCV BORDER_REFLECT_101; sobel(axis) = derivative along axis, [1 2 1] along the other
        sx = lambda a: ndi.sobel(a, axis=1, mode="mirror")
        sy = lambda a: ndi.sobel(a, axis=0, mode="mirror")
        assert np.allclose(Gxx, 0.125 * sx(Gx), atol=1e-14)
        assert np.allclose(Gyy, 0.125 * sy(Gy), atol=1e-14)
        assert np.allclose(Gxy, 0.5 * (0.125 * sy(Gx) + 0.125 * sx(Gy)), atol=1e-14)
    # known answer: a plane I = 3x + 5y has gradient (3,5) => d/dx of Gx=3 is 0; Gx = x gives Gxx = 1 away from borders
    X = np.tile(np.arange(10.0), (6, 1))
    Gxx, Gxy, Gyy = oracle_mod.hessian(X, X.copy())
    assert np.allclose(Gxx[:, 1:-1], 1.0) and np.allclose(Gyy, 0.0) and np.allclose(Gxy[:, 1:-1], 0.5)


def test_eval_data_error_quirks_and_invariants(oracle_mod):
    O = oracle_mod
    w = small_workload(n_events=20050)   # not a multiple of 100: the last 50 events are dropped (quirk Q1)
    r = oracle_run(O, w, dump=True, alpha=0)
    d, ep, nem = r["dump"], r["ep"], r["num_ev_map"]
    n = w.events.size()
    assert (d["inlier_idx"][20000:] == -2).all() and (d["cp_idx"][20000:] == -1).all()
    inl = d["inlier_idx"]
    assert ep.size == (inl >= 0).sum() == nem.sum()       # CHECK_GE + one count per inlier (model.cpp:227,247)
    # inlier indices enumerate 0..M-1 in sensor-pixel-major then time order (model.cpp:179-186)
    pix = w.events.y.astype(np.int64) * w.sensor_w + w.events.x
    order = np.lexsort((np.arange(n), pix))[np.isin(np.lexsort((np.arange(n), pix)), np.nonzero(inl >= 0)[0])]
    assert np.array_equal(inl[order], np.arange(ep.size))
    # outliers are exactly the measurements with |dp| > 10 px (model.cpp:200) or leaving the panorama (none here)
    meas = inl != -2
    nrm = np.hypot(d["dp"][:, 0], d["dp"][:, 1])
    assert ((nrm > 10) == (inl == -1))[meas].all()
    assert (inl == -1).sum() > 0 and (inl >= 0).sum() > 0
    # residual definition (model.cpp:217-221)
    k = np.nonzero(inl >= 0)[0]
    e = 2 * (w.events.polarity[k] - 0.5) * w.C_th - (d["Gpm"][k] * d["dp"][k]).sum(1)
    assert np.allclose(ep[inl[k]], e, rtol=0, atol=1e-15)
    # first event at each sensor pixel is not a measurement; predecessor is the previous event at the same pixel
    prev = d["prev"]
    has_prev = prev >= 0
    assert (pix[prev[has_prev]] == pix[has_prev]).all() and (prev[has_prev] < np.nonzero(has_prev)[0]).all()
    assert ((inl == -2)[:20000] == ~has_prev[:20000]).all()
    # pm_int is round-half-away-from-zero of pm (model.cpp:209-210)
    assert np.array_equal(d["pm_int"][k], np.floor(d["pm"][k] + 0.5).astype(np.int32))


def test_form_normal_eq_consistency_with_dense_jacobian(oracle_mod):
    """Hand-derived known answer: assemble the sparse Jacobian J (one row per active measurement) from the dumped state
    and check A = J^T J, b = J^T e block by block against formNormalEq (model.cpp:316-491)."""
    O = oracle_mod
    w = small_workload(n_events=6000, pano_h=128, K=5, sensor=(24, 16), focal=20.0)
    r = oracle_run(O, w, dump=True, alpha=0, dense_A12=True, thres=2)
    d, ne, ep, nem = r["dump"], r["ne"], r["ep"], r["num_ev_map"]
    P, K = ne["P"], w.K
    assert P > 10
    comp = -np.ones(w.pano_h * w.pano_w, dtype=np.int64); comp[ne["active"]] = np.arange(P)
    assert np.array_equal(ne["active"], np.nonzero(nem.ravel() >= 2)[0])   # ascending pano index (std::set order)
    rows = []
    for k in np.nonzero(d["inlier_idx"] >= 0)[0]:
        pi = d["pm_int"][k, 1] * w.pano_w + d["pm_int"][k, 0]
        if comp[pi] < 0:
            continue
        row = np.zeros(3 * K + 2 * P)
        kp = d["prev"][k]
        row[3 * d["cp_idx"][k]:3 * d["cp_idx"][k] + 6] += d["temp"][k] @ d["D"][k]
        row[3 * d["cp_idx"][kp]:3 * d["cp_idx"][kp] + 6] += -d["Gpm"][k] @ d["D"][kp]
        row[3 * K + 2 * comp[pi]:3 * K + 2 * comp[pi] + 2] = d["dp"][k]
        rows.append((row, ep[d["inlier_idx"][k]]))
    J = np.array([r_[0] for r_ in rows]); e = np.array([r_[1] for r_ in rows])
    A = J.T @ J; b = J.T @ e
    assert np.allclose(ne["A11"], A[:3 * K, :3 * K], rtol=1e-10, atol=1e-12)
    assert np.allclose(ne["A12"], A[:3 * K, 3 * K:], rtol=1e-10, atol=1e-12)
    assert np.allclose(ne["b1"], b[:3 * K], rtol=1e-10, atol=1e-12)
    assert np.allclose(ne["b2"], b[3 * K:], rtol=1e-10, atol=1e-12)
    for i in range(P):
        assert np.allclose(ne["A22"][i], A[3 * K + 2 * i:3 * K + 2 * i + 2, 3 * K + 2 * i:3 * K + 2 * i + 2], rtol=1e-10, atol=1e-12)


def test_irls_weights_and_l2_reg(oracle_mod):
    O = oracle_mod
    w = small_workload(n_events=6000, pano_h=128, K=5, sensor=(24, 16), focal=20.0)
    base = oracle_run(O, w, alpha=0, thres=2)
    # huber with a huge threshold == quadratic (all weights 1, model.cpp:609-612)
    hub = oracle_run(O, w, alpha=0, thres=2, irls=1, a=1e9)
    assert np.array_equal(hub["ne"]["A11"], base["ne"]["A11"]) and np.array_equal(hub["ne"]["b2"], base["ne"]["b2"])
    # cauchy with a -> 0 tends to quadratic; with a > 0 every weight is < 1 so the A22 diagonals shrink
    cau = oracle_run(O, w, alpha=0, thres=2, irls=2, a=5.0)
    assert (cau["ne"]["A22"][:, 0, 0] <= base["ne"]["A22"][:, 0, 0] + 1e-15).all()
    assert cau["ne"]["A22"][:, 0, 0].sum() < base["ne"]["A22"][:, 0, 0].sum()
    # applyL2Reg: A22 += alpha*I, b2 -= alpha*[Gx,Gy][active] (model.cpp:689-719)
    reg = oracle_run(O, w, alpha=3.0, thres=2)
    act = reg["ne"]["active"]
    assert np.allclose(reg["ne"]["A22"][:, 0, 0], base["ne"]["A22"][:, 0, 0] + 3.0)
    assert np.allclose(reg["ne"]["A22"][:, 0, 1], base["ne"]["A22"][:, 0, 1])
    assert np.allclose(reg["ne"]["b2"][0::2], base["ne"]["b2"][0::2] - 3.0 * w.Gx.ravel()[act])
    assert np.allclose(reg["ne"]["b2"][1::2], base["ne"]["b2"][1::2] - 3.0 * w.Gy.ravel()[act])
    # cost terms (solver.cpp:88-90, model.cpp:279-314)
    ep = base["ep"]
    assert O.data_cost(ep) == pytest.approx(0.5 * ep @ ep, rel=1e-12)
    assert O.data_cost(ep, 2, 0.1) == pytest.approx(0.5 / 0.1 * np.log1p(0.1 * ep * ep).sum(), rel=1e-12)
    hub_ref = np.where(np.abs(ep) < 0.1, 0.5 * ep * ep, 0.1 * np.abs(ep) - 0.005).sum()
    assert O.data_cost(ep, 1, 0.1) == pytest.approx(hub_ref, rel=1e-12)
    assert O.reg_cost(w.Gx, w.Gy, 5.0) == pytest.approx(2.5 * ((w.Gx ** 2).sum() + (w.Gy ** 2).sum()), rel=1e-12)


def test_empty_and_tiny_inputs(oracle_mod):
    O = oracle_mod
    w = small_workload(n_events=99)      # fewer than one batch: nothing is warped at all (quirk Q1)
    r = oracle_run(O, w, alpha=0)
    assert r["ep"].size == 0 and r["num_ev_map"].sum() == 0 and r["ne"]["P"] == 0
    assert not r["ne"]["A11"].any()
    w = small_workload(n_events=0)
    r = oracle_run(O, w, alpha=0)
    assert r["ep"].size == 0 and r["ne"]["P"] == 0


def test_update_map_known_answers(oracle_mod):
    """LEGM::updateMap (model.cpp:863-903): active += damping*x2 in ascending order, every other pixel := 0."""
    Gx = np.arange(12.0).reshape(3, 4) + 1; Gy = -Gx
    act = np.array([1, 5, 6], dtype=np.uint32); x2 = np.array([10.0, 20, 30, 40, 50, 60])
    nx, ny = oracle_mod.update_map(act, x2, 0.5, Gx, Gy)
    ex = np.zeros(12); ey = np.zeros(12)
    ex[[1, 5, 6]] = Gx.ravel()[[1, 5, 6]] + 0.5 * x2[0::2]; ey[[1, 5, 6]] = Gy.ravel()[[1, 5, 6]] + 0.5 * x2[1::2]
    assert np.array_equal(nx.ravel(), ex) and np.array_equal(ny.ravel(), ey)
    assert np.array_equal(Gx, np.arange(12.0).reshape(3, 4) + 1)          # inputs untouched (the reference clones first, solver.cpp:237)


def test_solve_normal_eq_solves_the_damped_system(oracle_mod):
    """LEGM::solveNormalEq (model.cpp:721-792): x = [x1; x2] must solve [A11m A12; A12^T A22m] x = [b1; b2]."""
    w = small_workload(n_events=6000, pano_h=128, K=5, sensor=(24, 16), focal=20.0)
    r = oracle_run(oracle_mod, w, dense_A12=True, thres=2)
    ne = r["ne"]
    for lam, fix in ((1e-3, False), (1.0, True)):
        x1, x2 = oracle_mod.solve_normal_eq(ne, lam, fix)
        K, P = w.K, ne["P"]
        A = np.zeros((3 * K + 2 * P, 3 * K + 2 * P)); b = np.concatenate([ne["b1"], ne["b2"]])
        A[:3 * K, :3 * K] = ne["A11"] + lam * np.diag(np.diag(ne["A11"]))
        A[:3 * K, 3 * K:] = ne["A12"]; A[3 * K:, :3 * K] = ne["A12"].T
        for i in range(P):
            blk = ne["A22"][i] + lam * np.diag(np.diag(ne["A22"][i]))
            A[3 * K + 2 * i:3 * K + 2 * i + 2, 3 * K + 2 * i:3 * K + 2 * i + 2] = blk
        keep = np.arange(3 if fix else 0, A.shape[0])
        x = np.linalg.solve(A[np.ix_(keep, keep)], b[keep])
        got = np.concatenate([x1, x2])[keep]
        assert np.allclose(got, x, rtol=1e-8, atol=1e-10 * np.abs(x).max())
        if fix:
            assert (x1[:3] == 0).all()


@pytest.mark.parametrize("irls,a,fix", [(0, 0.0, False), (1, 0.1, True), (2, 1.0, True)])
def test_sparse_oracle_solvers_match_the_dense_route(oracle_mod, irls, a, fix):
    """The oracle's Schur solve on the sparse A12 factors (what the full-size device tests are checked with) must agree with its
    dense restatement of LEGM::solveNormalEq (model.cpp:721-792); its restatement of solveNormalEqCG (model.cpp:794-840, Eigen's
    ConjugateGradient loop) must converge to the same solution when allowed to, and stop after max_iter like Eigen does."""
    from helpers import small_workload
    O = oracle_mod
    w = small_workload(n_events=20000)
    o = O.OracleLEGM(w.sensor_w, w.sensor_h, w.pano_w, w.pano_h, w.lut, w.C_th)
    ev = w.events
    ep, nem = o.evaluate_data_error(w.traj.knots_xyzw, w.traj.t0_ns, w.traj.dt_ns, w.Gx, w.Gy, ev.x, ev.y, ev.polarity, ev.t_ns)
    ne = o.apply_l2(o.form_normal_eq(ep, w.K, nem, 5, irls, a, True), w.alpha, w.Gx, w.Gy)
    for lam in (1e-3, 10.0):
        x1d, x2d = O.solve_normal_eq(ne, lam, fix)
        x1s, x2s = o.solve_sparse(ne, ep, w.K, nem, 5, irls, a, lam, fix)
        assert np.abs(x1s - x1d).max() <= 1e-9 * np.abs(x1d).max() and np.abs(x2s - x2d).max() <= 1e-9 * np.abs(x2d).max()
        x1c, x2c, it, err = o.solve_cg_sparse(ne, ep, w.K, nem, 5, irls, a, lam, fix, max_iter=5000, tol=1e-13)
        assert err < 1e-12 and np.abs(x1c - x1d).max() <= 1e-6 * np.abs(x1d).max() and np.abs(x2c - x2d).max() <= 1e-6 * np.abs(x2d).max()
    _, _, it, err = o.solve_cg_sparse(ne, ep, w.K, nem, 5, irls, a, 1e-3, fix)       # the reference's settings: 100 iterations, 1e-6
    assert it <= 100 and (it == 100 or err < 1e-6)
    if fix:
        assert np.all(x1s[:3] == 0) and np.all(x1c[:3] == 0)


def test_omp_mode_and_lean_count_map_agree_with_the_reference_order_pass(oracle_mod):
    """The "omp" CPU-baseline mode (bench.py) runs the same per-measurement arithmetic as the one-thread checker: identical ep
    (values and order), identical count map and active set, blocks equal to rounding; the lean index-level pass used for the
    flip-rate measurement gives the same count map, inlier count, pm and rounded pixels as the full pass."""
    from helpers import small_workload
    O = oracle_mod
    w = small_workload(n_events=30000)
    ev = w.events

    def run(threads):
        O.set_threads(threads)
        try:
            o = O.OracleLEGM(w.sensor_w, w.sensor_h, w.pano_w, w.pano_h, w.lut, w.C_th)
            ep, nem, d = o.evaluate_data_error(w.traj.knots_xyzw, w.traj.t0_ns, w.traj.dt_ns, w.Gx, w.Gy, ev.x, ev.y, ev.polarity, ev.t_ns, dump=True)
            ne = o.apply_l2(o.form_normal_eq(ep, w.K, nem, 5, 1, 0.1), w.alpha, w.Gx, w.Gy)
            lean = o.count_map(w.traj.knots_xyzw, w.traj.t0_ns, w.traj.dt_ns, ev.x, ev.y, ev.t_ns, want_pm=True, want_pm_int=True)
        finally:
            O.set_threads(1)
        return ep, nem, d, ne, lean

    ep1, nem1, d1, ne1, lean1 = run(1)
    ep4, nem4, d4, ne4, lean4 = run(4)
    assert np.array_equal(ep1, ep4) and np.array_equal(nem1, nem4) and np.array_equal(ne1["active"], ne4["active"])
    assert np.array_equal(d1["inlier_idx"], d4["inlier_idx"]) and np.array_equal(d1["pm"], d4["pm"])
    for k in ("A11", "b1", "A22", "b2"):
        assert np.abs(ne1[k] - ne4[k]).max() <= 1e-12 * np.abs(ne1[k]).max(), k
    for lean in (lean1, lean4):
        n_inl, nem_l, pm_l, pmi_l = lean
        n_used = (ev.size() // 100) * 100
        assert n_inl == ep1.size and np.array_equal(nem_l, nem1)
        assert np.array_equal(pm_l[:n_used], d1["pm"][:n_used]) and np.array_equal(pmi_l[:n_used], d1["pm_int"][:n_used])


# ---- f1: the solvers' Eigen calls, PINNED to the reference's own vendored Eigen (tests/golden/eigen_solvers.npz) ---------------------------
EIGEN_GOLD = os.path.join(os.path.dirname(__file__), "golden", "eigen_solvers.npz")


def test_ldlt_and_inverse2_match_the_reference_eigen_golden(oracle_mod):
    """S.ldlt().solve(rhs) (model.cpp:789) and Matrix2d::inverse() (model.cpp:750) as Eigen 3.3.9 executes them: the oracle's restatement
    takes the same pivots (transpositions identical), leaves the same D, gives the same solution — including a Schur complement with a
    control pose no event constrains (zero rows: Eigen pivots them last and its pseudo-inverse of D returns a ZERO update there), an
    indefinite matrix, the all-zero matrix and n = 1 — and the same inverse entries, inf / nan of a singular block included."""
    O = oracle_mod
    g = np.load(EIGEN_GOLD)
    for name in g["ldlt_names"]:
        S, rhs = g[f"ldlt_{name}_S"], g[f"ldlt_{name}_rhs"]
        x, d, tr, info = O.ldlt_solve(S, rhs)
        assert np.array_equal(tr, g[f"ldlt_{name}_tr"]), f"{name}: pivot order differs from Eigen's"
        assert info == int(g[f"ldlt_{name}_info"])
        ex, eD = g[f"ldlt_{name}_x"], g[f"ldlt_{name}_D"]
        assert np.array_equal(d == 0, eD == 0) and np.allclose(d, eD, rtol=1e-10, atol=1e-13 * np.abs(eD).max())
        assert np.array_equal(x == 0, ex == 0), f"{name}: zero components differ"
        assert np.allclose(x, ex, rtol=1e-9, atol=1e-11 * max(np.abs(ex).max(), 1e-300)), name
    x = O.ldlt_solve(g["ldlt_unobserved_pose_S"], g["ldlt_unobserved_pose_rhs"])[0]
    dead = np.diag(g["ldlt_unobserved_pose_S"]) == 0
    assert dead.sum() >= 3 and (x[dead] == 0).all() and (x[~dead] != 0).all()
    got = np.array([O.inverse2(a).ravel() for a in g["inv2_A"]])
    exp = g["inv2_out"]
    assert np.array_equal(np.isfinite(got), np.isfinite(exp)) and np.array_equal(np.isnan(got), np.isnan(exp))
    fin = np.isfinite(exp)
    assert np.array_equal(got[fin], exp[fin]), "2x2 inverse is expected to match Eigen bit for bit (same five operations)"
    assert np.array_equal(np.sign(got[np.isinf(exp)]), np.sign(exp[np.isinf(exp)]))


def test_cg_matches_the_reference_eigen_golden(oracle_mod):
    """solveNormalEqCG (model.cpp:794-840): the oracle's matrix-free restatement of Eigen's ConjugateGradient loop against Eigen's own run
    on the assembled sparse matrix — same number of iterations, same error estimate, same iterate."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from make_eigen_golden import full_system_triplets
    O = oracle_mod
    g = np.load(EIGEN_GOLD)
    w = small_workload(n_events=2400, pano_h=64, K=5, sensor=(12, 8), focal=10.0)
    r = oracle_run(O, w, dense_A12=True)
    ne, o = r["ne"], r["oracle"]
    for name in ("trimmed", "full"):
        lam, skip = float(g[f"cg_{name}_lam"]), int(g[f"cg_{name}_skip"])
        # the fixture's system was assembled by the REFERENCE's own eigen_utils::catSpMat / diagMat / diagSpMat (model.cpp:805-821, round 4) from
        # these blocks — which are the oracle's (the workload generator and the oracle are deterministic) ...
        assert np.allclose(ne["A11"][skip:, skip:], g[f"cg_{name}_A11"], rtol=1e-13, atol=0) and np.allclose(ne["A12"][skip:, :], g[f"cg_{name}_A12"], rtol=1e-13, atol=0)
        assert np.allclose(ne["A22"], g[f"cg_{name}_A22"], rtol=1e-13, atol=0) and np.allclose(ne["b2"], g[f"cg_{name}_b2"], rtol=1e-13, atol=0)
        # ... and the Python assembly of the same system (what the tests used before) is the same matrix
        n, rows, cols, vals, b = full_system_triplets(ne, lam, skip)
        M_py = np.zeros((n, n)); np.add.at(M_py, (rows, cols), vals)
        M_ref = np.zeros((n, n)); np.add.at(M_ref, (g[f"cg_{name}_rows"], g[f"cg_{name}_cols"]), g[f"cg_{name}_vals"])
        assert np.allclose(M_py, M_ref, rtol=1e-13, atol=0) and np.allclose(b, g[f"cg_{name}_b"], rtol=1e-13, atol=0)
        assert np.array_equal(M_ref, M_ref.T)
        x1, x2, it, err = o.solve_cg_sparse(ne, r["ep"], w.K, r["num_ev_map"], w.thres_valid_pixel, 0, 0.0, lam, bool(skip))
        ex = g[f"cg_{name}_x"]
        assert it == int(g[f"cg_{name}_iters"]), (it, int(g[f"cg_{name}_iters"]))
        assert err == pytest.approx(float(g[f"cg_{name}_err"]), rel=1e-6)
        got = np.concatenate([x1[skip:], x2])
        assert np.allclose(got, ex, rtol=1e-7, atol=1e-9 * np.abs(ex).max())


def test_solver_restatements_live_against_the_reference_eigen_build(oracle_mod):
    """Where oracle/_ref/libref_eigen.so is present (authoring container, and the GPU box: it travels prebuilt): random systems beyond the
    golden file — SPD, semi-definite with zero rows, indefinite — same pivots, same zero pattern, same solution."""
    O = oracle_mod
    if O.ref_eigen() is None:
        pytest.skip("oracle/_ref/libref_eigen.so not present")
    rng = np.random.default_rng(7)
    for trial in range(40):
        n = int(rng.integers(2, 60))
        B = rng.normal(size=(n, n))
        S = B @ B.T + 1e-2 * np.eye(n)
        rhs = rng.normal(size=n)
        kind = trial % 4
        if kind == 1:      # unobserved poses: whole rows / columns vanish
            dead = rng.choice(n, size=max(1, n // 5), replace=False)
            S[dead, :] = 0; S[:, dead] = 0; rhs[dead] = 0
        elif kind == 2:    # indefinite
            S[rng.integers(0, n), rng.integers(0, n)] *= -3; S = 0.5 * (S + S.T)
        elif kind == 3:    # equal diagonal entries: the pivot search must take the FIRST maximum
            S = np.eye(n) * 2.0 + 0.1 * (B + B.T) / n; np.fill_diagonal(S, 2.0)
        a, b = O.ldlt_solve(S, rhs), O.ref_ldlt_solve(S, rhs)
        assert np.array_equal(a[2], b[2]) and a[3] == b[3], f"trial {trial}: pivots / info"
        assert np.array_equal(a[0] == 0, b[0] == 0)
        assert np.allclose(a[0], b[0], rtol=1e-8, atol=1e-10 * np.abs(b[0]).max()), f"trial {trial}"
    for _ in range(100):
        A = rng.normal(size=4)
        assert np.array_equal(O.inverse2(A), O.ref_inverse2(A))
    # solveNormalEqCG end to end by the reference's own assembly code (eigen_utils.cpp) on a system that is not in the golden file
    w = small_workload(n_events=3000, pano_h=64, K=6, sensor=(12, 8), focal=10.0, seed=11)
    r = oracle_run(O, w, dense_A12=True)
    ne, o = r["ne"], r["oracle"]
    for lam, skip in ((1e-3, 3), (1e-1, 0)):
        x1r, x2r, itr, errr, _ = O.ref_solve_normal_eq_cg(ne["A11"][skip:, skip:], ne["A12"][skip:, :], ne["A22"], ne["b1"][skip:], ne["b2"], lam)
        x1, x2, it, err = o.solve_cg_sparse(ne, r["ep"], w.K, r["num_ev_map"], w.thres_valid_pixel, 0, 0.0, lam, bool(skip))
        assert it == itr and err == pytest.approx(errr, rel=1e-6)
        ex = np.concatenate([x1r, x2r]); got = np.concatenate([x1[skip:], x2])
        assert np.allclose(got, ex, rtol=1e-7, atol=1e-9 * np.abs(ex).max())
