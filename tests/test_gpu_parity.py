"""GPU parity tests (-m gpu): the HIP path, called through the C ABI (emba_amd.LEGM -> ctypes -> libemba_hip.so),
against the CPU oracle on identical seeded inputs.

Bar (BASELINE north_star): event->pixel indexing (pm_int, num_ev_map, active set, control-pose indices, inlier
numbering) BIT-EXACT; residuals / Jacobians / normal-equation blocks within 1e-5 relative (asserted, together with
a much tighter engineering bound of 1e-9, see helpers.py).
"""
import os
import numpy as np
import pytest

from helpers import assert_close, oracle_run, small_workload

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from emba_amd import build
    build.build_hip()
    return True


OPTIONS = {}      # emba_set_option values every context of a test gets (monkeypatch.setitem): the library's A/B switches are options, not env variables


def make_legm(w):
    from emba_amd import LEGM
    m = LEGM(w.sensor_w, w.sensor_h, w.lut, w.C_th, w.pano_w, w.pano_h, device=0)
    for k, v in OPTIONS.items():
        m.set_option(k, v)
    return m


def gpu_run(w, thres=None, cost_type="quadratic", a=0.0, alpha=None, dense_A12=False, dump=False):
    m = make_legm(w)
    nem = np.zeros((w.pano_h, w.pano_w), dtype=np.int32)
    ep = m.evaluateDataError(w.traj, w.Gx, w.Gy, w.events, True, nem)
    d = m.dump_state() if dump else None
    th = w.thres_valid_pixel if thres is None else thres
    if cost_type == "quadratic":
        ne = m.formNormalEq(ep, w.K, nem, th, dense_A12=dense_A12)
    else:
        ne = m.formNormalEqIRLS(ep, w.K, nem, th, cost_type, a, dense_A12=dense_A12)
    al = w.alpha if alpha is None else alpha
    if al:
        ne = m.applyL2Reg(al, dense_A12=dense_A12)
    return dict(ep=ep, num_ev_map=nem, ne=ne, dump=d, legm=m)


def compare_normal_eq(g, o, dense=False):
    assert g["P"] == o["P"]
    assert np.array_equal(g["active"], o["active"])            # bit-exact active set, ascending pano index
    errs = dict(A11=assert_close(g["A11"], o["A11"], "A11"), b1=assert_close(g["b1"], o["b1"], "b1"))
    if o["P"]:
        errs["A22"] = assert_close(g["A22"], o["A22"], "A22")
        errs["b2"] = assert_close(g["b2"], o["b2"], "b2")
        if dense:
            errs["A12"] = assert_close(g["A12"], o["A12"], "A12")
    return errs


def compare_event_state(gd, od, used):
    """Per-event State_LEGM (state.h:56-83): integers bit-exact, floats norm-wise to the engineering bounds AND element-wise to the contract's 1e-5."""
    assert np.array_equal(gd["cp_idx"], od["cp_idx"])
    assert np.array_equal(gd["inlier_idx"], od["inlier_idx"])
    assert np.array_equal(gd["pm_int"], od["pm_int"])
    assert_close(gd["pm"][:used], od["pm"][:used], "pm", tight=1e-12)
    assert_close(gd["D"][:used], od["D"][:used], "dpm_ddrot_cp", tight=1e-10)
    assert_close(gd["dp"], od["dp"], "dp", tight=1e-9)
    assert_close(gd["Gpm"], od["Gpm"], "Gpm", tight=1e-15)
    assert_close(gd["temp"], od["temp"], "temp", tight=1e-10)


def test_state_parity_per_event(gpu, oracle_mod):
    """Per-event State_LEGM after evaluateDataError: pm_int / cp_idx / inlier_idx bit-exact, floats within tolerance."""
    w = small_workload(n_events=20050)     # tail of 50 events dropped (quirk Q1)
    g = gpu_run(w, dump=True, alpha=0)
    o = oracle_run(oracle_mod, w, dump=True, alpha=0)
    gd, od = g["dump"], o["dump"]
    assert np.array_equal(gd["cp_idx"], od["cp_idx"])
    assert np.array_equal(gd["inlier_idx"], od["inlier_idx"])
    assert np.array_equal(gd["pm_int"], od["pm_int"])
    assert np.array_equal(g["num_ev_map"], o["num_ev_map"])
    used = 20000
    assert_close(gd["pm"][:used], od["pm"][:used], "pm", tight=1e-12)
    assert_close(gd["D"][:used], od["D"][:used], "dpm_ddrot_cp", tight=1e-10)
    assert_close(gd["dp"], od["dp"], "dp", tight=1e-9)
    assert_close(gd["Gpm"], od["Gpm"], "Gpm", tight=1e-15)
    assert_close(gd["temp"], od["temp"], "temp", tight=1e-10)
    assert g["ep"].shape == o["ep"].shape
    assert_close(g["ep"], o["ep"], "ep", tight=1e-9)


@pytest.mark.parametrize("cfg", [
    dict(n_events=20000),
    dict(n_events=50000, pano_h=512, K=11, sensor=(128, 128), focal=91.4015),      # playroom-like calib, 512x1024 pano
    dict(n_events=30000, pano_h=256, K=21, sensor=(64, 48), focal=60.0, dt_knots=0.01),  # many control poses, distant pairs
    dict(n_events=8000, pano_h=64, K=4, sensor=(16, 12), focal=12.0),              # heavy pixel collisions
])
def test_normal_equations_parity(gpu, oracle_mod, cfg):
    w = small_workload(**cfg)
    g = gpu_run(w, dense_A12=True)
    o = oracle_run(oracle_mod, w, dense_A12=True)
    assert np.array_equal(g["num_ev_map"], o["num_ev_map"])
    assert_close(g["ep"], o["ep"], "ep")
    errs = compare_normal_eq(g["ne"], o["ne"], dense=True)
    assert o["ne"]["P"] > 0 and o["ep"].size > 0, "degenerate test input"
    print(cfg, errs)


@pytest.mark.parametrize("cfg", [
    dict(n_events=20000),                                                                       # dense slots through the sparse form: several rounds of records per stage of 128 tags
    dict(n_events=30000, pano_h=256, K=21, sensor=(64, 48), focal=60.0, dt_knots=0.01),         # many control-pose pairs: stages that straddle pair boundaries
    dict(n_events=60000, pano_h=512, K=9, sensor=(64, 48), focal=120.0, thres_valid_pixel=6),   # a long focal length on a big panorama: few pixels reach the threshold — most slots dead
    dict(n_events=8000, pano_h=64, K=4, sensor=(16, 12), focal=12.0),
    "baseline",
])
@pytest.mark.parametrize("cost", [("quadratic", 0.0), ("huber", 0.1)])
def test_gram_sums_from_a_sparse_slot_stream(gpu, oracle_mod, cfg, cost, monkeypatch):
    """Round 5: the Gram kernel's form for slot streams with few live records (option gram_sparse = 1 forces it: stages of 128 tags, the live slots compacted
    through LDS, only their records fetched) gives the A11 | b1 of the dense form's — against the oracle, through formNormalEq[IRLS] and through two resident steps (the
    second chooses its form from the first one's active-pixel count when the option is left alone)."""
    from emba_amd.synth import make_workload
    monkeypatch.setitem(OPTIONS, "gram_sparse", 1)
    monkeypatch.setitem(OPTIONS, "order", 1)                 # pixel order: the tag stream exists there
    w = make_workload() if cfg == "baseline" else small_workload(**cfg)
    irls = {"quadratic": 0, "huber": 1}[cost[0]]
    g = gpu_run(w, cost_type=cost[0], a=cost[1])
    o = oracle_run(oracle_mod, w, irls=irls, a=cost[1])
    assert o["ne"]["P"] > 0
    compare_normal_eq(g["ne"], o["ne"])
    if irls == 0:
        m = make_legm(w)
        m.set_events(w.events); m.upload_map(w.Gx, w.Gy)
        for it in range(2):
            n_inl, P = m.step(w.traj, w.thres_valid_pixel, w.alpha)
            assert n_inl == o["ep"].size and P == o["ne"]["P"]
            compare_normal_eq(m._finish(w.alpha, False), o["ne"])


@pytest.mark.parametrize("cost_type,a,irls", [("huber", 0.1, 1), ("cauchy", 0.1, 2), ("cauchy", 5.0, 2), ("huber", 1e9, 1)])
def test_irls_parity(gpu, oracle_mod, cost_type, a, irls):
    w = small_workload(n_events=20000)
    g = gpu_run(w, cost_type=cost_type, a=a, dense_A12=True)
    o = oracle_run(oracle_mod, w, irls=irls, a=a, dense_A12=True)
    compare_normal_eq(g["ne"], o["ne"], dense=True)
    assert g["legm"].dataCost(cost_type, a) == pytest.approx(oracle_mod.data_cost(o["ep"], irls, a), rel=1e-10)


def test_costs_and_l2reg(gpu, oracle_mod):
    w = small_workload(n_events=20000)
    g = gpu_run(w, alpha=0)
    o = oracle_run(oracle_mod, w, alpha=0)
    m = g["legm"]
    assert m.dataCost() == pytest.approx(oracle_mod.data_cost(o["ep"]), rel=1e-10)
    assert m.regCost(5.0) == pytest.approx(oracle_mod.reg_cost(w.Gx, w.Gy, 5.0), rel=1e-10)
    compare_normal_eq(g["ne"], o["ne"])
    # applyL2Reg as a separate call after formNormalEq (solver.cpp:130)
    ne = m.applyL2Reg(3.0)
    oracle_mod_ne = o["oracle"].apply_l2(o["ne"], 3.0, w.Gx, w.Gy)
    compare_normal_eq(ne, oracle_mod_ne)


def test_external_ep_argument_is_used(gpu, oracle_mod):
    """formNormalEq takes `ep` from the caller (model.cpp:421): a modified vector must change b1/b2 accordingly."""
    w = small_workload(n_events=20000)
    m = make_legm(w)
    nem = np.zeros((w.pano_h, w.pano_w), dtype=np.int32)
    ep = m.evaluateDataError(w.traj, w.Gx, w.Gy, w.events, True, nem)
    o = oracle_run(oracle_mod, w, alpha=0)
    ep2 = o["ep"] * 0.5 + 0.01
    g = m.formNormalEq(ep2, w.K, nem, w.thres_valid_pixel)
    ref = o["oracle"].form_normal_eq(ep2, w.K, o["num_ev_map"], w.thres_valid_pixel)
    compare_normal_eq(g, ref)


def test_sparse_a12_reassembles_dense(gpu, oracle_mod):
    w = small_workload(n_events=20000)
    g = gpu_run(w, alpha=0)
    o = oracle_run(oracle_mod, w, alpha=0, dense_A12=True)
    s = g["legm"].A12_sparse()
    K, P = w.K, o["ne"]["P"]
    A12 = np.zeros((3 * K, 2 * P))
    ok = s["pix"] >= 0
    for i in range(6):
        for d in range(2):
            np.add.at(A12, (3 * s["cp_c"][ok] + i, 2 * s["pix"][ok] + d), s["w"][ok] * s["jc"][ok, i] * s["dp"][ok, d])
            np.add.at(A12, (3 * s["cp_p"][ok] + i, 2 * s["pix"][ok] + d), s["w"][ok] * s["jp"][ok, i] * s["dp"][ok, d])
    assert_close(A12, o["ne"]["A12"], "A12 from sparse factors")
    # records are sorted by control-pose pair
    key = (s["cp_c"].astype(np.int64) << 16) | s["cp_p"]
    assert (np.diff(key) >= 0).all()


def test_repeated_calls_and_new_trajectory(gpu, oracle_mod):
    """LM call pattern (solver.cpp:63-353): several evaluateDataError calls on one event set with changing poses/map."""
    from emba_amd.synth import so3_exp_xyzw
    w = small_workload(n_events=20000)
    m = make_legm(w)
    m.set_events(w.events)
    rng = np.random.default_rng(0)
    for it in range(3):
        nem = np.zeros((w.pano_h, w.pano_w), dtype=np.int32)
        ep = m.evaluateDataError(w.traj, w.Gx, w.Gy, None, True, nem)
        m.formNormalEq(ep, w.K, nem, w.thres_valid_pixel)
        ne = m.applyL2Reg(w.alpha)
        o = oracle_run(oracle_mod, w)
        assert np.array_equal(nem, o["num_ev_map"])
        assert_close(ep, o["ep"], "ep")
        compare_normal_eq(ne, o["ne"])
        # perturb poses (left multiply, trajectory.cpp:296-304) and the map, like an accepted LM step
        for i in range(w.K):
            e = so3_exp_xyzw(rng.normal(size=3) * 0.01)
            ex, ey, ez, ew = e; bx, by, bz, bw = w.traj.knots_xyzw[i]
            q = np.array([ew * bx + ex * bw + ey * bz - ez * by, ew * by + ey * bw + ez * bx - ex * bz,
                          ew * bz + ez * bw + ex * by - ey * bx, ew * bw - ex * bx - ey * by - ez * bz])
            w.traj.knots_xyzw[i] = q / np.linalg.norm(q)
        w.Gx = w.Gx + rng.normal(size=w.Gx.shape) * 1e-3
        w.Gy = w.Gy + rng.normal(size=w.Gy.shape) * 1e-3


def test_tile_order_rebins_after_trajectory_drift(gpu, oracle_mod, monkeypatch):
    """Tile order: the bins come from the first trajectory of a window.  A trajectory that has since moved most events out of their
    tile (+ 8-px margin) must still give the oracle's results (those events go to HBM one by one), is reported by
    emba_last_tile_drift, and makes the NEXT evaluation rebuild the order — after which (same poses) nothing is outside any more."""
    from emba_amd.synth import so3_exp_xyzw
    monkeypatch.setitem(OPTIONS, "order", 2)
    w = small_workload(n_events=60000, pano_h=256, K=11, sensor=(48, 36), focal=40.0)
    m = make_legm(w)
    m.set_events(w.events)

    def evaluate():
        nem = np.zeros((w.pano_h, w.pano_w), dtype=np.int32)
        ep = m.evaluateDataError(w.traj, w.Gx, w.Gy, None, True, nem)
        m.formNormalEq(ep, w.K, nem, w.thres_valid_pixel)
        ne = m.applyL2Reg(w.alpha)
        o = oracle_run(oracle_mod, w)
        assert np.array_equal(nem, o["num_ev_map"])
        assert_close(ep, o["ep"], "ep")
        compare_normal_eq(ne, o["ne"])

    evaluate()
    assert m.setup_info()["tile_order"]
    out0, rebins0 = m.tile_drift()
    assert out0 == 0 and rebins0 == 0
    # rotate every control pose by ~0.3 rad about the vertical axis: tens of panorama pixels at this focal length
    e = so3_exp_xyzw(np.array([0.0, 0.3, 0.0]))
    ex, ey, ez, ew = e
    for i in range(w.K):
        bx, by, bz, bw = w.traj.knots_xyzw[i]
        q = np.array([ew * bx + ex * bw + ey * bz - ez * by, ew * by + ey * bw + ez * bx - ex * bz,
                      ew * bz + ez * bw + ex * by - ey * bx, ew * bw - ex * bx - ey * by - ez * bz])
        w.traj.knots_xyzw[i] = q / np.linalg.norm(q)
    evaluate()                   # old bins, new poses: correct, but most inliers are found outside their tile
    out1, rebins1 = m.tile_drift()
    assert out1 > 0 and rebins1 == 1, (out1, rebins1)
    evaluate()                   # the order was rebuilt from these poses at the start of this evaluation
    out2, rebins2 = m.tile_drift()
    assert out2 == 0 and rebins2 == 1, (out2, rebins2)


def test_edge_cases(gpu, oracle_mod):
    from emba_amd import EmbaError
    # fewer events than one batch, and none at all
    for n in (0, 99):
        w = small_workload(n_events=n)
        g = gpu_run(w, alpha=0)
        assert g["ep"].size == 0 and g["num_ev_map"].sum() == 0 and g["ne"]["P"] == 0 and not g["ne"]["A11"].any()
    # threshold so high that no pixel is active
    w = small_workload(n_events=20000)
    g = gpu_run(w, thres=10**6, alpha=0)
    assert g["ne"]["P"] == 0 and not g["ne"]["A11"].any() and not g["ne"]["b1"].any()
    # every event on ONE sensor pixel (maximal collision), ragged size
    w = small_workload(n_events=1234)
    w.events.x[:] = 5; w.events.y[:] = 7
    g = gpu_run(w, thres=1, dense_A12=True)
    o = oracle_run(oracle_mod, w, thres=1, dense_A12=True)
    assert np.array_equal(g["num_ev_map"], o["num_ev_map"])
    assert_close(g["ep"], o["ep"], "ep")
    compare_normal_eq(g["ne"], o["ne"], dense=True)
    # a trajectory that does not cover the events: BASALT_ASSERT in the reference, EMBA_ERR_TIME_RANGE here
    w = small_workload(n_events=20000)
    m = make_legm(w)
    short = type(w.traj)(w.traj.knots_xyzw[:3].copy(), w.traj.t0_ns, w.traj.dt_ns)
    with pytest.raises(EmbaError) as ei:
        m.evaluateDataError(short, w.Gx, w.Gy, w.events, True, None)
    assert ei.value.status == 4
    # call-order violation: formNormalEq before evaluateDataError
    m2 = make_legm(w)
    m2.K = w.K
    with pytest.raises(EmbaError) as ei:
        m2.formNormalEq(None, w.K, None, 5)
    assert ei.value.status == 5
    # unsorted timestamps are rejected
    bad = small_workload(n_events=1000)
    bad.events.t_ns[10] = bad.events.t_ns[5]
    bad.events.t_ns[11] = bad.events.t_ns[3]
    with pytest.raises(EmbaError):
        make_legm(bad).set_events(bad.events)


def test_panorama_border_and_wraparound(gpu, oracle_mod):
    """Events looking backwards (phi near +-pi) and near the poles: pixels that round to W or H are outliers by definition
    (reference UB) on both sides; everything else must still match exactly."""
    from emba_amd.synth import so3_exp_xyzw
    w = small_workload(n_events=20000, pano_h=128)
    K = w.K
    w.traj.knots_xyzw[:] = np.stack([so3_exp_xyzw([0.0, np.pi - 0.25 + 0.12 * i, 0.0]) for i in range(K)])   # sweep across phi = pi
    g = gpu_run(w, thres=2, dump=True)
    o = oracle_run(oracle_mod, w, thres=2, dump=True)
    assert np.array_equal(g["dump"]["pm_int"], o["dump"]["pm_int"])
    assert np.array_equal(g["dump"]["inlier_idx"], o["dump"]["inlier_idx"])
    assert np.array_equal(g["num_ev_map"], o["num_ev_map"])
    assert (o["dump"]["pm_int"][:, 0].max() >= w.pano_w - 2) and (o["dump"]["pm_int"][o["dump"]["pm_int"][:, 0] >= 0, 0].min() <= 1), \
        "test input does not reach the panorama seam"
    compare_normal_eq(g["ne"], o["ne"])
    # pitch up to the pole
    w.traj.knots_xyzw[:] = np.stack([so3_exp_xyzw([1.2 + 0.08 * i, 0.0, 0.0]) for i in range(K)])
    g = gpu_run(w, thres=2, dump=True)
    o = oracle_run(oracle_mod, w, thres=2, dump=True)
    assert np.array_equal(g["dump"]["pm_int"], o["dump"]["pm_int"])
    assert np.array_equal(g["num_ev_map"], o["num_ev_map"])
    compare_normal_eq(g["ne"], o["ne"])


def test_full_size_properties(gpu):
    """BASELINE size (1 M events, 240x180 sensor, 1024x2048 panorama): size-independent properties instead of the oracle.
    (i) sum(num_ev_map) == #inliers == len(ep); (ii) active set == pixels with count >= thres; (iii) linearity: doubling C_th
    and the map doubles ep and b, quadruples nothing in A11 but scales it as expected; (iv) trace(A22) == sum_w |dp|^2 over
    active measurements recomputed from the sparse factors; (v) idempotence: a second identical call returns identical integers
    and floats equal to rounding."""
    from emba_amd.synth import make_workload
    from emba_amd import LEGM
    w = make_workload()      # BASELINE configuration
    m = LEGM(w.sensor_w, w.sensor_h, w.lut, w.C_th, w.pano_w, w.pano_h)
    nem = np.zeros((w.pano_h, w.pano_w), dtype=np.int32)
    ep = m.evaluateDataError(w.traj, w.Gx, w.Gy, w.events, True, nem)
    ne = m.formNormalEq(ep, w.K, nem, w.thres_valid_pixel)
    assert nem.sum() == ep.size and 0.5 * w.events.size() < ep.size < w.events.size()
    assert np.array_equal(ne["active"], np.nonzero(nem.ravel() >= w.thres_valid_pixel)[0])
    assert np.allclose(ne["A11"], ne["A11"].T, rtol=1e-12, atol=1e-9)
    s = m.A12_sparse()
    ok = s["pix"] >= 0
    assert ok.sum() == nem.ravel()[ne["active"]].sum()            # every inlier at an active pixel contributes once
    tr = np.bincount(s["pix"][ok], weights=(s["dp"][ok] ** 2).sum(1), minlength=ne["P"])
    assert np.allclose(ne["A22"][:, 0, 0] + ne["A22"][:, 1, 1], tr, rtol=1e-10)
    # A11 / b1 rebuilt from the sparse factors, one 12x12 Gram matrix per control-pose pair
    A11 = np.zeros_like(ne["A11"]); b1 = np.zeros_like(ne["b1"])
    order = {int(i): r for r, i in enumerate(ne["active"])}
    v = np.concatenate([s["jc"][ok], s["jp"][ok]], axis=1)
    key = (s["cp_c"][ok].astype(np.int64) << 16) | s["cp_p"][ok]
    # residual per record: ep is in reference order, so recover e from b2 instead: use the identity b1 = sum v*e via A12^T? keep A11 only
    for k in np.unique(key):
        sel = key == k
        c, pp = int(k >> 16), int(k & 0xFFFF)
        M = v[sel].T @ v[sel]
        idx = np.r_[3 * c:3 * c + 6, 3 * pp:3 * pp + 6]
        np.add.at(A11, (idx[:, None], idx[None, :]), M)
    assert np.allclose(ne["A11"], A11, rtol=1e-9, atol=1e-10 * np.abs(A11).max())
    del order, b1
    assert m.dataCost() == pytest.approx(0.5 * ep @ ep, rel=1e-10)
    # idempotence
    nem2 = np.zeros_like(nem)
    ep2 = m.evaluateDataError(w.traj, w.Gx, w.Gy, None, True, nem2)
    ne2 = m.formNormalEq(ep2, w.K, nem2, w.thres_valid_pixel)
    assert np.array_equal(nem, nem2) and np.array_equal(ep, ep2) and np.array_equal(ne["active"], ne2["active"])
    assert np.allclose(ne["A11"], ne2["A11"], rtol=1e-11, atol=1e-9 * np.abs(ne["A11"]).max())
    # linearity in (C_th, map): e -> 2e, jc/jp -> 2x, dp unchanged  =>  A11 -> 4x, b1 -> 4x, A22 same, b2 -> 2x
    m2 = LEGM(w.sensor_w, w.sensor_h, w.lut, 2 * w.C_th, w.pano_w, w.pano_h)
    nem3 = np.zeros_like(nem)
    ep3 = m2.evaluateDataError(w.traj, 2 * w.Gx, 2 * w.Gy, w.events, True, nem3)
    ne3 = m2.formNormalEq(ep3, w.K, nem3, w.thres_valid_pixel)
    assert np.array_equal(nem3, nem)
    assert np.allclose(ep3, 2 * ep, rtol=1e-12, atol=1e-15)
    assert np.allclose(ne3["A11"], 4 * ne["A11"], rtol=1e-9, atol=1e-9 * np.abs(ne["A11"]).max())
    assert np.allclose(ne3["b2"], 2 * ne["b2"], rtol=1e-9, atol=1e-12)
    assert np.allclose(ne3["A22"], ne["A22"], rtol=1e-10)


def test_baseline_size_against_oracle(gpu, oracle_mod):
    """The BASELINE configuration itself (1 M events, 240x180, 1024x2048, K=21) against the oracle: the CPU side takes about a
    second, so the full-size check does not have to rely on properties alone."""
    from emba_amd.synth import make_workload
    w = make_workload()
    g = gpu_run(w, dump=True)
    o = oracle_run(oracle_mod, w, dump=True)
    assert np.array_equal(g["num_ev_map"], o["num_ev_map"])            # 675 197 rounded pixels, bit-exact
    assert g["ep"].shape == o["ep"].shape
    assert_close(g["ep"], o["ep"], "ep")
    errs = compare_normal_eq(g["ne"], o["ne"])
    print("baseline-size parity:", errs)
    # round 6: the per-event state at this size too — every event's 2 x 6 Jacobian, displacement and temp row, value by value
    compare_event_state(g["dump"], o["dump"], w.events.size() // 100 * 100)
    assert g["legm"].dataCost() == pytest.approx(oracle_mod.data_cost(o["ep"]), rel=1e-11)


@pytest.mark.parametrize("cfg", [
    dict(n_events=1500000, pano_h=2048, K=256, sensor=(640, 480), focal=520.0, dt_knots=0.004, thres_valid_pixel=2),   # city/synthetic-like: big sensor, big pano, many knots
    dict(n_events=1000000, pano_h=512, K=47, sensor=(128, 128), focal=91.4015, dt_knots=0.05, t_beg=0.1),  # config 1: playroom.launch on calib/DVS-playroom.yaml (128x128, f = 91.4015), 2.3-s window, ~1 M events
    dict(n_events=400000, pano_h=256, K=6, sensor=(32, 24), focal=30.0),                         # dense stream: ~500 events per sensor pixel
    dict(n_events=600000, pano_h=1024, K=201, sensor=(240, 180), focal=200.0, dt_knots=0.05, thres_valid_pixel=2),   # shapes.launch shape: 10 s window, 5 rad sweep (footprint as wide as the panorama)
])
def test_other_configurations_against_oracle(gpu, oracle_mod, cfg):
    from emba_amd.synth import make_workload
    w = make_workload(**cfg)
    g = gpu_run(w, dump=True)
    o = oracle_run(oracle_mod, w, dump=True)
    assert np.array_equal(g["num_ev_map"], o["num_ev_map"])
    assert_close(g["ep"], o["ep"], "ep")
    compare_normal_eq(g["ne"], o["ne"])
    compare_event_state(g["dump"], o["dump"], w.events.size() // 100 * 100)       # (config 1 among them: every event's state, value by value)
    assert o["ep"].size > 10000 and o["ne"]["P"] > 100, "degenerate test input"


@pytest.mark.parametrize("mode", ["fly", "pack", "rect"])
def test_hessian_sources_agree(gpu, oracle_mod, mode, monkeypatch):
    """The three ways the warp kernel can obtain the Hessian (3x3 stencil on the planes, full texel pack, texel rectangle of the
    previous footprint) must give identical results; run two evaluations so that the rectangle is populated."""
    monkeypatch.setitem(OPTIONS, "texel", {"pack": 1, "fly": 2, "rect": 3}[mode])
    w = small_workload(n_events=30000)
    m = make_legm(w)
    m.set_events(w.events)
    o = oracle_run(oracle_mod, w)
    for it in range(3):
        nem = np.zeros((w.pano_h, w.pano_w), dtype=np.int32)
        ep = m.evaluateDataError(w.traj, w.Gx, w.Gy, None, True, nem)
        m.formNormalEq(ep, w.K, nem, w.thres_valid_pixel)
        ne = m.applyL2Reg(w.alpha)
        assert np.array_equal(nem, o["num_ev_map"])
        assert_close(ep, o["ep"], "ep")
        compare_normal_eq(ne, o["ne"])


def test_lm_loop_map_residency(gpu, oracle_mod):
    """the map stays on the device across LM trial points.  updateMap (model.cpp:863-903) builds the trial map,
    evaluateDataError(traj, None, None) runs on it, accept/reject follow the LM decision (solver.cpp:299-352)."""
    from emba_amd import EmbaError
    w = small_workload(n_events=20000)
    m = make_legm(w)
    m.set_events(w.events)
    rng = np.random.default_rng(1)
    Gx, Gy = w.Gx.copy(), w.Gy.copy()                       # oracle-side current map
    nem = np.zeros((w.pano_h, w.pano_w), dtype=np.int32)
    ep0 = m.evaluateDataError(w.traj, Gx, Gy, None, True, nem)   # uploads once
    with pytest.raises(EmbaError):
        m.acceptMap()                                        # nothing to accept yet
    for step, decision in enumerate(["reject", "accept", "accept"]):
        m.formNormalEq(ep0, w.K, nem, w.thres_valid_pixel)
        ne = m.applyL2Reg(w.alpha)
        x2 = rng.normal(size=2 * ne["P"]) * 0.01
        m.updateMap(x2, 0.7)
        Gx_t, Gy_t = oracle_mod.update_map(ne["active"], x2, 0.7, Gx, Gy)
        dx, dy = m.downloadMap()
        assert np.allclose(dx, Gx_t, rtol=0, atol=1e-16) and np.allclose(dy, Gy_t, rtol=0, atol=1e-16)
        assert (dx.ravel()[np.setdiff1d(np.arange(dx.size), ne["active"])] == 0).all()
        # trial evaluation on the resident map == oracle on the updated map
        wt = small_workload(n_events=20000); wt.Gx, wt.Gy = Gx_t, Gy_t
        ot = oracle_run(oracle_mod, wt)
        nem_t = np.zeros_like(nem)
        ep_t = m.evaluateDataError(w.traj, None, None, None, True, nem_t)
        assert np.array_equal(nem_t, ot["num_ev_map"])
        assert_close(ep_t, ot["ep"], "trial ep")
        assert m.regCost(w.alpha) == pytest.approx(oracle_mod.reg_cost(Gx_t, Gy_t, w.alpha), rel=1e-12)
        if decision == "reject":
            m.rejectMap()
            nem_b = np.zeros_like(nem)
            ep_b = m.evaluateDataError(w.traj, None, None, None, True, nem_b)
            assert np.array_equal(nem_b, nem) and np.array_equal(ep_b, ep0)     # back on the old map, bit for bit
        else:
            m.acceptMap()
            Gx, Gy, ep0, nem = Gx_t, Gy_t, ep_t, nem_t
            ne2 = m.formNormalEq(ep0, w.K, nem, w.thres_valid_pixel)
            compare_normal_eq(m.applyL2Reg(w.alpha), ot["ne"])
            nem_c = np.zeros_like(nem)
            ep0 = m.evaluateDataError(w.traj, None, None, None, True, nem_c)     # state for the next formNormalEq
            assert np.array_equal(nem_c, nem)


@pytest.mark.parametrize("cfg,lam,fix,cost", [
    (dict(n_events=20000), 1e-3, True, ("quadratic", 0.0)),
    (dict(n_events=20000), 10.0, False, ("quadratic", 0.0)),
    (dict(n_events=30000, pano_h=256, K=21, sensor=(64, 48), focal=60.0, dt_knots=0.01), 1e-2, True, ("huber", 0.1)),
    (dict(n_events=8000, pano_h=64, K=4, sensor=(16, 12), focal=12.0), 1e-3, True, ("cauchy", 1.0)),
    # 3K - 3 = 219 unknown pose rows: four 64-wide tiles, so the multi-tile SYRK, the panel TRSM and the trailing updates of the
    # blocked Cholesky all run (K <= 22 fits one tile)
    (dict(n_events=60000, pano_h=256, K=74, sensor=(64, 48), focal=60.0, dt_knots=0.004, thres_valid_pixel=3), 1e-2, True, ("quadratic", 0.0)),
    (dict(n_events=40000, pano_h=128, K=45, sensor=(32, 24), focal=30.0, dt_knots=0.01, thres_valid_pixel=3), 1e-1, False, ("huber", 0.1)),
])
def test_schur_solve_parity(gpu, oracle_mod, cfg, lam, fix, cost):
    """LEGM::solveNormalEq (model.cpp:721-792) on the device from the sparse A12 factors, against the dense CPU
    restatement; also the residual of the full damped system."""
    w = small_workload(**cfg)
    irls = {"quadratic": 0, "huber": 1, "cauchy": 2}[cost[0]]
    g = gpu_run(w, cost_type=cost[0], a=cost[1])
    o = oracle_run(oracle_mod, w, irls=irls, a=cost[1], dense_A12=True)
    x1, x2 = g["legm"].solveNormalEq(lam, fix_first_pose=fix)
    ox1, ox2 = oracle_mod.solve_normal_eq(o["ne"], lam, fix)
    assert x1.shape == ox1.shape and x2.shape == ox2.shape
    assert np.allclose(x1, ox1, rtol=1e-7, atol=1e-9 * np.abs(ox1).max())
    assert np.allclose(x2, ox2, rtol=1e-7, atol=1e-9 * np.abs(ox2).max())
    if fix:
        assert (x1[:3] == 0).all()
    # calling it again (e.g. a rejected step retried with a larger lambda, solver.cpp:181-202) works on the same state
    x1b, x2b = g["legm"].solveNormalEq(10 * lam, fix_first_pose=fix)
    ob1, ob2 = oracle_mod.solve_normal_eq(o["ne"], 10 * lam, fix)
    assert np.allclose(x1b, ob1, rtol=1e-7, atol=1e-9 * np.abs(ob1).max()) and np.allclose(x2b, ob2, rtol=1e-7, atol=1e-9 * np.abs(ob2).max())


def _unobserved_pose_workload(K=8, frac=0.55, **kw):
    """K control poses, events only in the first part of the window: the last control poses are constrained by no event (their rows and
    columns of A11, A12 — hence of the Schur complement — are exactly zero)."""
    from emba_amd.legm import EventPacket
    cfg = dict(n_events=20000, pano_h=128, K=K, sensor=(32, 24), focal=30.0)
    cfg.update(kw)
    w = small_workload(**cfg)
    ev = w.events
    n = (int(ev.size() * frac) // 100) * 100
    w.events = EventPacket(ev.x[:n], ev.y[:n], ev.polarity[:n], ev.t_ns[:n])
    return w


@pytest.mark.parametrize("K,fix", [(22, True), (22, False), (23, True), (43, False), (44, True), (65, True), (65, False)])
def test_schur_solve_at_the_panel_edges_of_the_factorisation(gpu, oracle_mod, K, fix):
    """The blocked Cholesky's shapes around its 64-wide panels (round 4: single-launch form for m + 1 <= 64, trailing update + next diagonal block in
    one launch): m = 3K - 3 or 3K unknown pose rows = 63 (the single-launch form, full), 66 (a panel and a 2-row one), 129 (two panels and a 1-row one),
    192 (three full panels: the last trailing tile holds nothing but the right-hand-side row), 195."""
    w = small_workload(n_events=40000, pano_h=128, K=K, sensor=(32, 24), focal=30.0, dt_knots=0.01, thres_valid_pixel=3)
    g = gpu_run(w)
    o = oracle_run(oracle_mod, w, dense_A12=True)
    for lam in (1e-2, 3.0):
        x1, x2 = g["legm"].solveNormalEq(lam, fix_first_pose=fix)
        ox1, ox2 = oracle_mod.solve_normal_eq(o["ne"], lam, fix)
        assert np.allclose(x1, ox1, rtol=1e-7, atol=1e-9 * np.abs(ox1).max())
        assert np.allclose(x2, ox2, rtol=1e-7, atol=1e-9 * np.abs(ox2).max())


@pytest.mark.parametrize("perm", ["0", "1"])
def test_schur_solve_with_the_columns_of_U_in_panorama_column_order(gpu, oracle_mod, perm, monkeypatch):
    """Round 4: the local Schur solve orders U's columns by panorama column when the camera pans (solve_perm: fewer (row-block pair, slice) products in the
    block-sparse SYRK).  Any order gives the same S, x1, x2: forced on and off (option solve_perm) at K = 100 — five row blocks, block-sparse form — against the
    oracle, then a re-solve with another lambda on the cached lists + order."""
    monkeypatch.setitem(OPTIONS, "solve_perm", int(perm))
    w = small_workload(n_events=120000, pano_h=256, K=100, sensor=(64, 48), focal=60.0, dt_knots=0.004, thres_valid_pixel=3)
    g = gpu_run(w)
    assert g["ne"]["P"] >= 512              # (below four slices the order is not used)
    o = oracle_run(oracle_mod, w, dense_A12=True)
    for lam in (1e-2, 1.0):
        x1, x2 = g["legm"].solveNormalEq(lam, fix_first_pose=True)
        ox1, ox2 = oracle_mod.solve_normal_eq(o["ne"], lam, True)
        assert np.allclose(x1, ox1, rtol=1e-7, atol=1e-9 * np.abs(ox1).max())
        assert np.allclose(x2, ox2, rtol=1e-7, atol=1e-9 * np.abs(ox2).max())


@pytest.mark.parametrize("opts", [dict(syrk_lists=1), dict(syrk_lists=2), dict(syrk_lists=2, syrk_item_cap=8), dict(syrk_dense=1), dict(syrk_dense=1, syrk_min_cols=64),
                                  dict(syrk_lists=2, solve_perm=1), dict(syrk_lists=1, solve_perm=0)])
def test_schur_solve_under_every_form_of_the_syrk(gpu, oracle_mod, opts, monkeypatch):
    """the block-sparse SYRK has three forms — per-pair slice LISTS, (pair, slice chunk) ITEMS (three list kernels, the slab reduce, and the
    overflow branch where an item beyond the slabs adds to S by global atomics) — and the dense split-K form; which one runs is decided from the band of U, so a
    long window is the only thing that reaches the item form by itself.  Every form forced (options syrk_lists / syrk_dense / syrk_min_cols / syrk_item_cap:
    8 slabs for hundreds of items) at K = 100 (five row blocks) against the oracle, with a re-solve on the cached lists."""
    for k, v in opts.items():
        monkeypatch.setitem(OPTIONS, k, v)
    w = small_workload(n_events=120000, pano_h=256, K=100, sensor=(64, 48), focal=60.0, dt_knots=0.004, thres_valid_pixel=3)
    g = gpu_run(w)
    assert g["ne"]["P"] >= 512
    o = oracle_run(oracle_mod, w, dense_A12=True)
    for lam in (1e-2, 1.0):
        x1, x2 = g["legm"].solveNormalEq(lam, fix_first_pose=True)
        ox1, ox2 = oracle_mod.solve_normal_eq(o["ne"], lam, True)
        assert np.allclose(x1, ox1, rtol=1e-7, atol=1e-9 * np.abs(ox1).max()), opts
        assert np.allclose(x2, ox2, rtol=1e-7, atol=1e-9 * np.abs(ox2).max()), opts


@pytest.mark.parametrize("K,kw", [(8, {}), (90, dict(n_events=60000, pano_h=256, sensor=(64, 48), focal=60.0, dt_knots=0.004, thres_valid_pixel=3))])
def test_schur_solve_with_an_unobserved_control_pose(gpu, oracle_mod, K, kw):
    """model.cpp:789 is Eigen's PIVOTED ldlt: for a semi-definite S (a control pose no event constrains) it returns, with a ZERO update in
    the unconstrained components (pinned against the reference's Eigen: tests/golden/eigen_solvers.npz).  The device factorisation must do
    the same — not raise, not produce NaN — and agree with the oracle everywhere else (one 64-wide tile and several)."""
    w = _unobserved_pose_workload(K, **kw)
    g = gpu_run(w)
    o = oracle_run(oracle_mod, w, dense_A12=True)
    dead = np.diag(o["ne"]["A11"]) == 0
    assert dead.sum() >= 3 and not dead[:3 * (K // 2)].any()
    for lam, fix in ((1e-3, True), (1e-1, False)):
        x1, x2 = g["legm"].solveNormalEq(lam, fix_first_pose=fix)
        ox1, ox2 = oracle_mod.solve_normal_eq(o["ne"], lam, fix)
        assert np.isfinite(x1).all() and np.isfinite(x2).all()
        assert (x1[dead] == 0).all() and (ox1[dead] == 0).all(), "unconstrained control poses must get a zero update"
        assert np.allclose(x1, ox1, rtol=1e-7, atol=1e-9 * np.abs(ox1).max()) and np.allclose(x2, ox2, rtol=1e-7, atol=1e-9 * np.abs(ox2).max())
        assert g["legm"].last_solve_info() == 2          # a pivot vanished (diagnostic), no 2x2 block failed
    # and the LM loop walks through it like the reference would: same decisions as the oracle loop, no "reject forever"
    from emba_amd import LEGM
    from emba_amd.solver import BASettings, LMSettings, solve_time_window
    from helpers import OracleModel
    from test_lm_solver_cpu import perturbed
    if K > 8:
        return
    init = perturbed(w, 0.003)
    ba, lm = BASettings(alpha=5.0), LMSettings(max_num_iter=4)
    m = LEGM(w.sensor_w, w.sensor_h, w.lut, w.C_th, w.pano_w, w.pano_h, device=0)
    rg = solve_time_window(m, init, w.events, w.Gx, w.Gy, ba, lm, resident=True)
    ro = solve_time_window(OracleModel(oracle_mod, w), init, w.events, w.Gx, w.Gy, ba, lm)
    assert [e[4] for e in rg.log] == [e[4] for e in ro.log] and any(e[4] for e in rg.log)
    for a, b in zip(rg.log, ro.log):
        assert a[3] == pytest.approx(b[3], rel=1e-7)
    assert np.abs(rg.traj.knots_xyzw - ro.traj.knots_xyzw).max() < 1e-8
    assert np.array_equal(rg.traj.knots_xyzw[-1], init.knots_xyzw[-1])      # the unconstrained last pose never moved


@pytest.mark.parametrize("cfg,lam,fix,cost", [
    (dict(n_events=20000), 1e-3, True, ("quadratic", 0.0)),
    (dict(n_events=30000, pano_h=256, K=21, sensor=(64, 48), focal=60.0, dt_knots=0.01), 1e-2, False, ("huber", 0.1)),
])
def test_cg_solve_parity(gpu, oracle_mod, cfg, lam, fix, cost):
    """LEGM::solveNormalEqCG (model.cpp:794-840) on the device against the oracle's restatement of Eigen's ConjugateGradient loop: with the
    reference's settings (100 iterations, 1e-6) the same stopping behaviour and iterates that agree to the solver's own tolerance; run to
    convergence, the solution of the direct Schur solve."""
    w = small_workload(**cfg)
    irls = {"quadratic": 0, "huber": 1, "cauchy": 2}[cost[0]]
    g = gpu_run(w, cost_type=cost[0], a=cost[1])
    o = oracle_run(oracle_mod, w, irls=irls, a=cost[1])
    m, orc = g["legm"], o["oracle"]
    thres = w.thres_valid_pixel
    x1, x2, it, err = m.solveNormalEqCG(lam, fix_first_pose=fix)
    ox1, ox2, oit, oerr = orc.solve_cg_sparse(o["ne"], o["ep"], w.K, o["num_ev_map"], thres, irls, cost[1], lam, fix)
    # the residual norm crosses the threshold within an iteration or two of the oracle's (different summation order), both below 1e-6
    assert abs(it - oit) <= 2 and it < 100 and err < 1e-6 and oerr < 1e-6
    assert np.abs(x1 - ox1).max() <= 1e-4 * np.abs(ox1).max() and np.abs(x2 - ox2).max() <= 1e-4 * np.abs(ox2).max()
    # same number of iterations -> the same iterate to rounding
    x1k, x2k, itk, _ = m.solveNormalEqCG(lam, fix_first_pose=fix, max_iter=10, tol=1e-30)
    o1k, o2k, oitk, _ = orc.solve_cg_sparse(o["ne"], o["ep"], w.K, o["num_ev_map"], thres, irls, cost[1], lam, fix, max_iter=10, tol=1e-30)
    assert itk == oitk == 10
    assert np.abs(x1k - o1k).max() <= 1e-8 * np.abs(o1k).max() and np.abs(x2k - o2k).max() <= 1e-8 * np.abs(o2k).max()
    if fix:
        assert (x1[:3] == 0).all()
    # to convergence: the direct solution
    x1c, x2c, itc, errc = m.solveNormalEqCG(lam, fix_first_pose=fix, max_iter=5000, tol=1e-13)
    d1, d2 = m.solveNormalEq(lam, fix_first_pose=fix)
    assert errc < 1e-12 and np.abs(x1c - d1).max() <= 1e-6 * np.abs(d1).max() and np.abs(x2c - d2).max() <= 1e-6 * np.abs(d2).max()


def test_lm_iterations_decrease_cost(gpu, oracle_mod):
    """A few full Levenberg-Marquardt iterations entirely through the device path (evaluate -> form -> L2 -> solve -> update map and
    poses -> evaluate), following solver.cpp:63-353: with a good damping the accepted steps must lower the total cost."""
    from emba_amd.synth import so3_exp_xyzw
    w = small_workload(n_events=30000)
    m = make_legm(w)
    m.set_events(w.events)
    nem = np.zeros((w.pano_h, w.pano_w), dtype=np.int32)
    traj = type(w.traj)(w.traj.knots_xyzw.copy(), w.traj.t0_ns, w.traj.dt_ns)
    ep = m.evaluateDataError(traj, w.Gx, w.Gy, None, True, nem)
    cost = m.dataCost() + m.regCost(w.alpha)
    lam, accepted = 1e-2, 0
    for it in range(6):
        m.formNormalEq(ep, w.K, nem, w.thres_valid_pixel)
        m.applyL2Reg(w.alpha)
        x1, x2 = m.solveNormalEq(lam, fix_first_pose=True)
        knots_new = traj.knots_xyzw.copy()
        for i in range(1, w.K):                           # left perturbation exp(x1_i) * knot_i, first pose fixed (trajectory.cpp:296-304)
            e = so3_exp_xyzw(x1[3 * i:3 * i + 3])
            ex, ey, ez, ew = e; bx, by, bz, bw = knots_new[i]
            q = np.array([ew * bx + ex * bw + ey * bz - ez * by, ew * by + ey * bw + ez * bx - ex * bz,
                          ew * bz + ez * bw + ex * by - ey * bx, ew * bw - ex * bx - ey * by - ez * bz])
            knots_new[i] = q / np.linalg.norm(q)
        m.updateMap(x2, 1.0)
        trial = type(traj)(knots_new, traj.t0_ns, traj.dt_ns)
        nem_new = np.zeros_like(nem)
        ep_new = m.evaluateDataError(trial, None, None, None, True, nem_new)
        cost_new = m.dataCost() + m.regCost(w.alpha)
        if cost_new < cost:
            m.acceptMap(); traj, ep, nem, cost, lam, accepted = trial, ep_new, nem_new, cost_new, lam / 10, accepted + 1
        else:
            m.rejectMap(); lam *= 10
            ep = m.evaluateDataError(traj, None, None, None, True, nem)    # restore the state formNormalEq reads
    assert accepted >= 2, "LM made no progress"


@pytest.mark.parametrize("cost", [("quadratic", 0.0), ("huber", 0.1)])
def test_single_call_step_matches_oracle(gpu, oracle_mod, cost):
    """emba_step (what bench.py times on one GPU: the whole resident step in one call, applyL2Reg folded into the active-set
    gather for the quadratic cost) must leave exactly the normal equations of the phase-by-phase path."""
    w = small_workload(n_events=30000)
    m = make_legm(w)
    m.set_events(w.events)
    m.upload_map(w.Gx, w.Gy)
    irls = {"quadratic": 0, "huber": 1}[cost[0]]
    o = oracle_run(oracle_mod, w, irls=irls, a=cost[1])
    for it in range(3):
        n_inl, P = m.step(w.traj, w.thres_valid_pixel, w.alpha, cost[0], cost[1])
        assert n_inl == o["ep"].size and P == o["ne"]["P"]
        ne = m._finish(w.alpha, False)                     # download only: L2 was already applied inside the step
        compare_normal_eq(ne, o["ne"])
        _, ep, nem = m.eval_finish(want_ep=True, want_map=True)
        assert np.array_equal(nem, o["num_ev_map"])
        assert_close(ep, o["ep"], "ep")
    x1, x2 = m.solveNormalEq(1e-2, fix_first_pose=True)    # and the solve works on that state
    o2 = oracle_run(oracle_mod, w, irls=irls, a=cost[1], dense_A12=True)
    ox1, ox2 = oracle_mod.solve_normal_eq(o2["ne"], 1e-2, True)
    assert np.allclose(x1, ox1, rtol=1e-7, atol=1e-9 * np.abs(ox1).max()) and np.allclose(x2, ox2, rtol=1e-7, atol=1e-9 * np.abs(ox2).max())


@pytest.mark.parametrize("resident", [False, True])
@pytest.mark.parametrize("ba_kw", [dict(alpha=0.0), dict(alpha=5.0), dict(use_IRLS=True, cost_type="huber", eta=0.1, alpha=1.0)])
def test_lm_solver_device_matches_oracle_loop(gpu, oracle_mod, ba_kw, resident):
    """EMBA::solveTimeWindow (solver.cpp:11-368) end to end — emba_amd.solver drives the device-resident path (hot path
    + f1 solve + f2 map) on events simulated from a scene; the same loop on the CPU oracle must take the same accept/reject
    decisions, reach the same costs and the same refined trajectory and map."""
    from emba_amd import synth
    from emba_amd.solver import BASettings, LMSettings, solve_time_window
    from helpers import OracleModel
    from test_lm_solver_cpu import knot_errors, perturbed
    w = synth.make_scene_workload(n_steps=1000)
    init = perturbed(w)
    ba, lm = BASettings(**ba_kw), LMSettings(max_num_iter=12)
    om = OracleModel(oracle_mod, w)
    ro = solve_time_window(om, init, w.events, w.Gx, w.Gy, ba, lm)
    m = make_legm(w)
    rg = solve_time_window(m, init, w.events, w.Gx, w.Gy, ba, lm, resident=resident)
    assert [e[4] for e in rg.log] == [e[4] for e in ro.log], "accept/reject sequence differs"
    assert rg.iterations == ro.iterations and rg.converged == ro.converged
    for g, o in zip(rg.log, ro.log):
        assert g[3] == pytest.approx(o[3], rel=1e-7) and g[2] == pytest.approx(o[2], rel=1e-7)
    assert np.abs(rg.traj.knots_xyzw - ro.traj.knots_xyzw).max() < 1e-7
    for d, o in zip(m.downloadMap(), om.downloadMap()):
        assert np.abs(d - o).max() < 1e-7 * np.abs(o).max()
    assert rg.cost_min < rg.log[0][2]
    if ba.alpha == 0.0:
        assert knot_errors(rg.traj, w.traj).mean() < knot_errors(init, w.traj).mean()


@pytest.mark.parametrize("pano_h", [75, 256, 1024, 2048])
def test_poisson_reconstruction_matches_oracle(gpu, pano_h, monkeypatch):
    """reconstructFromGradient (poisson_reconstruction.cpp:9-50, laplace.cpp:587-797) against the numpy restatement: Fourier
    analysis along H (sine-matrix products on the fp64 matrix cores, folded by the matrix's symmetry for even H) + tridiagonal solves along W;
    75 x 150 exercises the ragged tiles, the unaligned loads and the unfolded odd length, 1024 x 2048 is the BASELINE panorama, 2048 x 4096
    config 5's.  The round-1/2 form (both axes by sine-matrix products, option poisson = 1) must give the same panorama."""
    from oracle import poisson as OP
    w = small_workload(n_events=2000, pano_h=pano_h)
    m = make_legm(w)
    rng = np.random.default_rng(pano_h)
    Gx, Gy = rng.normal(size=(w.pano_h, w.pano_w)), rng.normal(size=(w.pano_h, w.pano_w))
    M = m.reconstructIntensity(Gx, Gy)
    Mo = OP.reconstruct_from_gradient(Gx, Gy)
    assert_close(M, Mo, "intensity panorama", tight=1e-10)
    # the device result solves the discrete equation it is defined by
    P = np.pad(M, 1)
    lap = P[2:, 1:-1] + P[:-2, 1:-1] + P[1:-1, 2:] + P[1:-1, :-2] - 4.0 * M
    assert np.abs(lap - OP.divergence(Gx, Gy)).max() < 1e-9 * np.abs(Gx).max() * w.pano_w
    # resident map: same result as passing it
    m.upload_map(Gx, Gy)
    assert np.array_equal(m.reconstructIntensity(), M)
    if pano_h <= 256:
        m.set_option("poisson", 1)
        assert_close(m.reconstructIntensity(), Mo, "intensity panorama (dense form)", tight=1e-10)


@pytest.mark.parametrize("declared,used", [(("huber", 0.1), ("huber", 0.1)), (("huber", 0.1), ("cauchy", 1.0)), (("cauchy", 1.0), ("quadratic", 0.0)),
                                           (("quadratic", 0.0), ("huber", 0.1)), (("huber", 0.1), ("huber", 0.2))])
def test_declared_cost_is_a_pure_speed_hint(gpu, oracle_mod, declared, used):
    """emba_set_cost lets the evaluation accumulate IRLS-weighted A22/b2 directly (model.cpp:599-636); whatever was declared,
    formNormalEq[IRLS] must return the normal equations of the cost it is CALLED with (a mismatch falls back to the records)."""
    w = small_workload(n_events=30000)
    m = make_legm(w)
    m.set_cost(*declared)
    nem = np.zeros((w.pano_h, w.pano_w), dtype=np.int32)
    ep = m.evaluateDataError(w.traj, w.Gx, w.Gy, w.events, True, nem)
    if used[0] == "quadratic":
        m.formNormalEq(ep, w.K, nem, w.thres_valid_pixel)
    else:
        m.formNormalEqIRLS(ep, w.K, nem, w.thres_valid_pixel, used[0], used[1])
    ne = m.applyL2Reg(w.alpha)
    irls = {"quadratic": 0, "huber": 1, "cauchy": 2}[used[0]]
    o = oracle_run(oracle_mod, w, irls=irls, a=used[1])
    assert np.array_equal(nem, o["num_ev_map"])
    compare_normal_eq(ne, o["ne"])


@pytest.mark.parametrize("sharded", [False, True])
def test_run_ba_command_line_on_files(gpu, tmp_path, sharded):
    """examples/run_ba.py on data from disk: event file, pose file, Gx.bin/Gy.bin, calibration -> refined trajectory,
    refined map and the Poisson-reconstructed panorama, in the reference's formats."""
    import subprocess, sys, os
    from emba_amd import io as eio, synth
    w = synth.make_scene_workload(n_steps=800)
    eio.save_events(tmp_path / "ev.npz", w.events)
    eio.save_map(tmp_path / "map", w.Gx, w.Gy)
    # initial poses: the ground-truth spline sampled densely (the front-end's trajectory), in the pose-file format
    from emba_amd import so3
    t0, dt, K = w.traj.t0_ns * 1e-9, w.traj.dt_ns * 1e-9, w.K
    with open(tmp_path / "poses.txt", "w") as f:
        for t in np.linspace(t0, t0 + dt * (K - 1), 120):
            i = min(int((t - t0) / dt), K - 2); u = (t - t0 - i * dt) / dt
            q = so3.mul(w.traj.knots_xyzw[i], so3.exp(u * so3.log(so3.mul(so3.inverse(w.traj.knots_xyzw[i]), w.traj.knots_xyzw[i + 1]))))
            f.write("%.9f 0 0 0 %.17g %.17g %.17g %.17g\n" % (t, q[0], q[1], q[2], q[3]))
    np.savez(tmp_path / "calib.npz", K=np.array([[60.0, 0, 32.0], [0, 60.0, 24.0], [0, 0, 1.0]]), D=np.zeros(5), width=64, height=48)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = tmp_path / "out"
    r = subprocess.run([sys.executable, os.path.join(root, "examples", "run_ba.py"), str(out), "--events", str(tmp_path / "ev.npz"),
                        "--poses", str(tmp_path / "poses.txt"), "--map-dir", str(tmp_path / "map"), "--calib", str(tmp_path / "calib.npz"),
                        "--dt-knots", "0.05", "--t-beg", "0.1", "--t-end", "0.35", "--alpha", "0.0", "--max-iter", "10"]
                       + (["--sharded"] if sharded else []),      # the torchrun host with one rank: ShardedModel, RCCL collectives at world size 1
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    t, qs = eio.load_poses(out / "refined_traj.txt")
    assert qs.shape == (6, 4) and np.allclose(t, 0.1 + 0.05 * np.arange(6), atol=1e-6)
    gx, gy = eio.load_map(out)
    assert gx.shape == w.Gx.shape and np.isfinite(gx).all()
    assert (out / "map_poisson_opt.pgm").stat().st_size > w.pano_h * w.pano_w


def test_randomised_small_configurations(gpu, oracle_mod):
    """A seeded sweep over odd shapes — sensor and panorama sizes that are not multiples of anything, K from 2 up, event counts with
    ragged tails, every threshold/cost — through the one-shot interface, the solve and a second evaluation on the same context
    (so the lazily cleared per-pixel state and the texel rectangle of the previous footprint are exercised as well)."""
    rng = np.random.default_rng(20240907)
    for case in range(14):
        sw, sh = int(rng.integers(5, 70)), int(rng.integers(5, 50))
        pano_h = int(rng.integers(20, 200))
        K = int(rng.integers(2, 12))
        n = int(rng.integers(100, 30000))
        cost = [("quadratic", 0.0), ("huber", 0.1), ("cauchy", 1.0)][case % 3]
        thres = int(rng.integers(1, 5))
        w = small_workload(n_events=n, pano_h=pano_h, K=K, sensor=(sw, sh), focal=float(rng.uniform(0.6, 1.5) * sw), seed=100 + case,
                           dt_knots=float(rng.choice([0.01, 0.05])), thres_valid_pixel=thres, alpha=float(rng.choice([0.0, 5.0])))
        irls = {"quadratic": 0, "huber": 1, "cauchy": 2}[cost[0]]
        tag = f"case {case}: {w.describe()} cost={cost} thres={thres} alpha={w.alpha}"
        o = oracle_run(oracle_mod, w, irls=irls, a=cost[1], dense_A12=True)
        g = gpu_run(w, cost_type=cost[0], a=cost[1], dense_A12=True)
        assert np.array_equal(g["num_ev_map"], o["num_ev_map"]), tag
        assert g["ep"].shape == o["ep"].shape, tag
        if o["ep"].size:
            assert_close(g["ep"], o["ep"], "ep " + tag)
        compare_normal_eq(g["ne"], o["ne"], dense=True)
        m = g["legm"]
        if o["ne"]["P"] and K >= 2:
            try:
                ox1, ox2 = oracle_mod.solve_normal_eq(o["ne"], 1e-2, True)
            except ValueError:
                ox1 = None
            if ox1 is not None and K > 1 and np.isfinite(ox1).all():
                x1, x2 = m.solveNormalEq(1e-2, fix_first_pose=True)
                assert np.allclose(x1, ox1, rtol=1e-6, atol=1e-8 * max(np.abs(ox1).max(), 1e-30)), tag
                assert np.allclose(x2, ox2, rtol=1e-6, atol=1e-8 * max(np.abs(ox2).max(), 1e-30)), tag
        # second evaluation on the same context with a slightly different trajectory
        knots = w.traj.knots_xyzw.copy(); knots[-1] = knots[-1] + 1e-3; knots[-1] /= np.linalg.norm(knots[-1])
        w.traj = type(w.traj)(knots, w.traj.t0_ns, w.traj.dt_ns)
        o2 = oracle_run(oracle_mod, w, irls=irls, a=cost[1])
        nem = np.zeros((w.pano_h, w.pano_w), dtype=np.int32)
        ep2 = m.evaluateDataError(w.traj, None, None, None, True, nem)
        assert np.array_equal(nem, o2["num_ev_map"]), tag + " (second evaluation)"
        if o2["ep"].size:
            assert_close(ep2, o2["ep"], "ep2 " + tag)


@pytest.mark.skipif(not os.environ.get("EMBA_SOAK"), reason="opt-in soak: EMBA_SOAK=<cases> (a one-off sweep after kernel changes, minutes)")
def test_randomised_soak(gpu, oracle_mod):
    """EMBA_SOAK=N: N more seeded random configurations than test_randomised_small_configurations, with wider ranges (K up to 45: the multi-panel
    factorisation; up to 150 k events), through evaluation, normal equations, solve, a second evaluation + formation and two resident single-call steps on the same context."""
    n_cases = int(os.environ["EMBA_SOAK"])
    if os.environ.get("EMBA_SOAK_ORDER"):      # "tile": the second sweep, with the tile order forced (option order = 2)
        OPTIONS["order"] = {"pixel": 1, "tile": 2}[os.environ["EMBA_SOAK_ORDER"]]
    rng = np.random.default_rng(777)
    bad = []
    for case in range(n_cases):
        sw, sh = int(rng.integers(5, 90)), int(rng.integers(5, 70))
        pano_h = int(rng.integers(20, 300))
        K = int(rng.integers(2, 46))
        n = int(rng.integers(100, 150000))
        cost = [("quadratic", 0.0), ("huber", 0.1), ("cauchy", 1.0)][case % 3]
        thres = int(rng.integers(1, 6))
        w = small_workload(n_events=n, pano_h=pano_h, K=K, sensor=(sw, sh), focal=float(rng.uniform(0.6, 1.5) * sw), seed=5000 + case,
                           dt_knots=float(rng.choice([0.004, 0.01, 0.05])), thres_valid_pixel=thres, alpha=float(rng.choice([0.0, 5.0])))
        irls = {"quadratic": 0, "huber": 1, "cauchy": 2}[cost[0]]
        tag = f"case {case}: {w.describe()} cost={cost} thres={thres} alpha={w.alpha}"
        if case < int(os.environ.get("EMBA_SOAK_FIRST", "0")):
            continue
        # every third case in pixel order through the Gram kernel's sparse-stream form, every third in tile order (round 5; options change speed only)
        OPTIONS.clear()
        if case % 3 == 1:
            OPTIONS.update(gram_sparse=1, order=1)
        elif case % 3 == 2:
            # (round 6) ... with every LDS tile shape, both origin grids and reserves 0 ... 4 of the window rule (each shape is a kernel instantiation of its own)
            OPTIONS.update(order=2, tile_shape=int(rng.integers(-1, 4)), tile_fine=int(rng.integers(-1, 2)), tile_reserve=int(rng.integers(0, 5)))
        tag += f" options={dict(OPTIONS)}"
        try:
            o = oracle_run(oracle_mod, w, irls=irls, a=cost[1], dense_A12=True)
            g = gpu_run(w, cost_type=cost[0], a=cost[1])
            assert np.array_equal(g["num_ev_map"], o["num_ev_map"]), "count map"
            assert g["ep"].shape == o["ep"].shape, "ep shape"
            if o["ep"].size:
                assert_close(g["ep"], o["ep"], "ep")
            compare_normal_eq(g["ne"], o["ne"])
            m = g["legm"]
            if o["ne"]["P"]:
                try:
                    ox1, ox2 = oracle_mod.solve_normal_eq(o["ne"], 1e-2, True)
                except ValueError:
                    ox1 = None
                # (a control pose that next to nothing constrains makes S singular to rounding — alpha = 0, a handful of events in its interval: the
                # oracle's LDLT then returns 1e13-sized components that no other factorisation reproduces; such systems are not compared)
                if ox1 is not None and np.isfinite(ox1).all() and np.abs(ox1).max() < 1e3:
                    try:
                        x1, x2 = m.solveNormalEq(1e-2, fix_first_pose=True)
                    except Exception as e:   # noqa: BLE001
                        # alpha = 0 and a pixel whose measurements all share one direction: A22_i is singular, the reference's inverse() gives
                        # inf / nan (model.cpp:750) — EMBA_ERR_NUMERIC is the documented answer (solver.py rejects the step, as the reference would)
                        if not (w.alpha == 0.0 and "EMBA_ERR_NUMERIC" in str(e)):
                            raise
                        x1 = None
                    if x1 is not None:
                        assert np.allclose(x1, ox1, rtol=1e-6, atol=1e-8 * max(np.abs(ox1).max(), 1e-30)), "x1"
                        assert np.allclose(x2, ox2, rtol=1e-6, atol=1e-8 * max(np.abs(ox2).max(), 1e-30)), "x2"
            knots = w.traj.knots_xyzw.copy(); knots[-1] = knots[-1] + 1e-3; knots[-1] /= np.linalg.norm(knots[-1])
            w.traj = type(w.traj)(knots, w.traj.t0_ns, w.traj.dt_ns)
            o2 = oracle_run(oracle_mod, w, irls=irls, a=cost[1])
            nem = np.zeros((w.pano_h, w.pano_w), dtype=np.int32)
            ep2 = m.evaluateDataError(w.traj, None, None, None, True, nem)
            assert np.array_equal(nem, o2["num_ev_map"]), "count map (second evaluation)"
            if o2["ep"].size:
                assert_close(ep2, o2["ep"], "ep2")
            ne2 = m.formNormalEq(ep2, w.K, nem, thres) if cost[0] == "quadratic" else m.formNormalEqIRLS(ep2, w.K, nem, thres, cost[0], cost[1])
            if w.alpha:
                ne2 = m.applyL2Reg(w.alpha)
            compare_normal_eq(ne2, o2["ne"])
            # the resident single-call step (what bench.py times) twice on the same context, at the second trajectory
            for _ in range(2):
                n_inl, P = m.step(w.traj, thres, w.alpha, cost[0], cost[1])
                assert n_inl == o2["ep"].size and P == o2["ne"]["P"], "step counts"
                compare_normal_eq(m._finish(w.alpha, False), o2["ne"])
            m.close()
        except AssertionError as e:
            bad.append(f"{tag}: {e}")
        except Exception as e:   # noqa: BLE001 (a status from the library: reported with its case, the sweep goes on)
            bad.append(f"{tag}: {type(e).__name__}: {e}")
    OPTIONS.clear()
    assert not bad, "\n".join(bad[:10])


@pytest.mark.parametrize("use_cg", [False, True])
def test_update_map_from_the_solvers_resident_x2(gpu, oracle_mod, use_cg):
    """solver.cpp:193-239 hands x2 from solveNormalEq[CG] straight to updateMap: with resident_x2 the 2P doubles stay on the device
    (solve returns None, updateMap(None) applies the device copy).  Same trial map as through the host, bit for bit; a NULL x2 with no
    solve of the CURRENT equations behind it is refused; the device-pointer form (a sharded host's all-reduced x2) gives the same map."""
    import torch
    from emba_amd import EmbaError
    w = small_workload(n_events=20000)
    m = make_legm(w)
    nem = np.zeros((w.pano_h, w.pano_w), dtype=np.int32)
    m.evaluateDataError(w.traj, w.Gx, w.Gy, w.events, True, nem)
    m.formNormalEq(None, w.K, nem, w.thres_valid_pixel)
    ne = m.applyL2Reg(w.alpha)
    with pytest.raises(EmbaError):
        m.updateMap(None, 0.7)                               # nothing solved yet
    solve = (lambda **kw: m.solveNormalEqCG(1e-2, True, **kw)[:2]) if use_cg else (lambda **kw: m.solveNormalEq(1e-2, True, **kw))
    x1h, x2h = solve()
    m.updateMap(x2h, 0.7)
    via_host = m.downloadMap()
    m.rejectMap()
    x1r, x2r = solve(resident_x2=True)
    # (two solves of the same system agree to rounding only: the product's partial tiles are combined with LDS atomics)
    # (... and two CG runs to the solver's own tolerance, model.cpp:830-836: the matrix is applied with atomics)
    rel = 1e-4 if use_cg else 1e-9
    assert x2r is None and np.allclose(x1r, x1h, rtol=rel, atol=rel * np.abs(x1h).max())
    m.updateMap(None, 0.7)
    via_dev = m.downloadMap()
    tol = rel * np.abs(x2h).max()
    assert np.allclose(via_dev[0], via_host[0], rtol=0, atol=tol) and np.allclose(via_dev[1], via_host[1], rtol=0, atol=tol)
    assert not np.array_equal(via_dev[0], w.Gx)              # (the trial map really is a different map)
    Gx_t, Gy_t = oracle_mod.update_map(ne["active"], x2h, 0.7, w.Gx, w.Gy)
    assert np.allclose(via_dev[0], Gx_t, rtol=0, atol=tol) and np.allclose(via_dev[1], Gy_t, rtol=0, atol=tol)
    m.rejectMap()
    t = torch.from_numpy(x2h).to("cuda:0")
    torch.cuda.synchronize()
    m.updateMap(int(t.data_ptr()), 0.7)
    via_ptr = m.downloadMap()
    assert np.array_equal(via_ptr[0], via_host[0]) and np.array_equal(via_ptr[1], via_host[1])
    m.rejectMap()
    # new equations: what the previous solve left on the device is not theirs
    m.evaluateDataError(w.traj, None, None, None, True, nem)
    m.formNormalEq(None, w.K, nem, w.thres_valid_pixel)
    m.applyL2Reg(w.alpha)
    with pytest.raises(EmbaError):
        m.updateMap(None, 0.7)


@pytest.mark.parametrize("cost", [("quadratic", 0.0), ("huber", 0.1), ("cauchy", 1.0)])
def test_costs_in_one_call(gpu, oracle_mod, cost):
    """emba_costs = (emba_data_cost, emba_reg_cost) of the same state with one host synchronisation (solver.cpp:88-91, 265-268)."""
    w = small_workload(n_events=20000)
    m = make_legm(w)
    nem = np.zeros((w.pano_h, w.pano_w), dtype=np.int32)
    ep = m.evaluateDataError(w.traj, w.Gx, w.Gy, w.events, True, nem)
    d, r = m.costs(cost[0], cost[1], w.alpha)
    assert d == pytest.approx(m.dataCost(*cost), rel=1e-12) and r == pytest.approx(m.regCost(w.alpha), rel=1e-12)      # (different reduction trees: the two forms use different grids)
    irls = {"quadratic": 0, "huber": 1, "cauchy": 2}[cost[0]]
    assert d == pytest.approx(oracle_mod.data_cost(ep, irls, cost[1]), rel=1e-10)
    assert r == pytest.approx(oracle_mod.reg_cost(w.Gx, w.Gy, w.alpha), rel=1e-12)


@pytest.mark.parametrize("cfg", [
    dict(n_events=30000),                                       # small: 64 units of 2048 pixels over 27 Gram blocks
    dict(n_events=20000, pano_h=75, sensor=(48, 36), focal=45.0),   # 75 x 150 panorama: 11 250 pixels, neither a multiple of 2048 nor of 8
    dict(n_events=60000, pano_h=256, K=9, sensor=(64, 48), focal=60.0, thres_valid_pixel=2),
    "baseline",                                                 # 1 M events, 1024 x 2048, K = 21: what bench.py steps
])
@pytest.mark.parametrize("fast,gather", [("1", "2"), ("1", "1"), ("1", "0"), ("0", "2"), ("0", "0")])
def test_resident_step_sequences(gpu, oracle_mod, cfg, fast, gather, monkeypatch):
    """The resident one-GPU step (emba_step) in the sequences a host produces.  Round 4: its active-set write + A22 | b2 gather is list-driven and rides in the head
    of the Gram kernel (option step_gather = 2; 1: a kernel of its own; 0: the sweeping kernel of the other paths), the accumulator lines are zeroed
    behind their readers so that the next evaluation has no clearing pass (option step_fast = 0 keeps the pass), and the count map's entries are stamped instead of cleared — each of which could leak one evaluation's state into the
    next.  Checked against the oracle after: two steps in a row; an evaluation that is never formed (a rejected trial) in between; a second
    formNormalEq, with another threshold, on the evaluation a step has consumed (A22 | b2 then come from the records)."""
    from emba_amd.synth import make_workload
    monkeypatch.setitem(OPTIONS, "step_fast", int(fast)); monkeypatch.setitem(OPTIONS, "step_gather", int(gather))
    w = make_workload() if cfg == "baseline" else small_workload(**cfg)
    m = make_legm(w)
    m.set_events(w.events)
    m.upload_map(w.Gx, w.Gy)
    o = oracle_run(oracle_mod, w)

    def check(tag):
        ne = m._finish(w.alpha, False)                     # download only: L2 was applied inside the step
        compare_normal_eq(ne, o["ne"])
        _, ep, nem = m.eval_finish(want_ep=True, want_map=True)
        assert np.array_equal(nem, o["num_ev_map"]), tag
        assert_close(ep, o["ep"], "ep " + tag)

    for it in range(2):
        n_inl, P = m.step(w.traj, w.thres_valid_pixel, w.alpha)
        assert n_inl == o["ep"].size and P == o["ne"]["P"]
        check(f"step {it}")
    # an evaluation at other poses that nobody forms: its markers and per-pixel sums must not reach the next step
    import copy
    traj2 = copy.deepcopy(w.traj)
    k = traj2.knots_xyzw.copy(); k[:, 0] += 0.01; k /= np.linalg.norm(k, axis=1, keepdims=True); traj2.knots_xyzw = k
    m.eval_launch(traj2); m.eval_finish(sync=False)
    assert m.dataCost() > 0
    m.eval_launch(traj2); m.eval_finish(sync=False)           # twice: markers of two stamps nobody materialised
    n_inl, P = m.step(w.traj, w.thres_valid_pixel, w.alpha)
    assert n_inl == o["ep"].size and P == o["ne"]["P"]
    check("step after unformed evaluations")
    # a second formNormalEq on the evaluation the step has consumed, with a lower threshold and without the L2 term
    th2 = max(1, w.thres_valid_pixel - 1)
    o2 = oracle_run(oracle_mod, w, thres=th2, alpha=0.0)
    ne2 = m.formNormalEq(None, w.K, None, th2)             # (ep = None: the device-resident residuals, like solver.cpp's call pattern)
    compare_normal_eq(ne2, o2["ne"])
    # ... and the step after that is clean again
    n_inl, P = m.step(w.traj, w.thres_valid_pixel, w.alpha)
    assert P == o["ne"]["P"]
    check("step after a re-form")


@pytest.mark.parametrize("cfg", [
    dict(n_events=30000),                                                         # 30 000 pm entries: 8 tail blocks, the last one ragged
    dict(n_events=4100, pano_h=75, sensor=(48, 36), focal=45.0),                  # just past one tail block
    dict(n_events=20300, pano_h=128, K=5, sensor=(32, 24), focal=30.0, thres_valid_pixel=2),
    dict(n_events=20000, order=2),                                                # tile order: flags / residuals are indexed in pm-order there too
    "baseline",
])
def test_resident_step_returns_ep_in_reference_order(gpu, oracle_mod, cfg, monkeypatch):
    """Round 5: the resident step produces what evaluateDataError RETURNS — the inliers' residuals in the reference's order
    (model.cpp:221,256: sensor pixel major, then time) — in tail blocks of its Gram launch, without a scan or compaction launch.  The vector the
    step left on the device must be the oracle's ep element for element, step after step, with and without unformed evaluations in between; with
    the option off (step_ep = 0) the stand-alone compaction gives the same vector."""
    from emba_amd.synth import make_workload
    if cfg != "baseline" and "order" in cfg:
        cfg = dict(cfg); monkeypatch.setitem(OPTIONS, "order", cfg.pop("order"))
    w = make_workload() if cfg == "baseline" else small_workload(**cfg)
    m = make_legm(w)
    m.set_events(w.events)
    m.upload_map(w.Gx, w.Gy)
    o = oracle_run(oracle_mod, w)
    assert o["ep"].size > 100
    for it in range(2):
        n_inl, P = m.step(w.traj, w.thres_valid_pixel, w.alpha)
        assert n_inl == o["ep"].size and P == o["ne"]["P"]
        ep = m.get_ep()
        assert ep.shape == o["ep"].shape
        assert_close(ep, o["ep"], f"ep of step {it}")
    m.eval_launch(w.traj); m.eval_finish(sync=False)              # an evaluation nobody forms, then a step again
    n_inl, P = m.step(w.traj, w.thres_valid_pixel, w.alpha)
    assert_close(m.get_ep(), o["ep"], "ep after an unformed evaluation")
    compare_normal_eq(m._finish(w.alpha, False), o["ne"])
    m.set_option("step_ep", 0)
    m.step(w.traj, w.thres_valid_pixel, w.alpha)
    assert m.get_option("ep_valid") == 0
    assert_close(m.get_ep(), o["ep"], "ep on demand")
    m.set_option("step_ep", 2)                                    # the form windows of more than 8.4 M entries get: scan + compaction launches behind the Gram launch
    m.step(w.traj, w.thres_valid_pixel, w.alpha)
    assert m.get_option("ep_valid") == 1
    assert_close(m.get_ep(), o["ep"], "ep by the launches behind the Gram kernel")
    m.set_option("step_ep", 1)
    m.step(w.traj, w.thres_valid_pixel, w.alpha, "huber", 0.1)   # IRLS step: same residual vector (the weights are formNormalEq's)
    assert_close(m.get_ep(), o["ep"], "ep of an IRLS step")


def test_empty_window_after_a_resident_step(gpu, oracle_mod):
    """the count map is stamped, not cleared, between evaluations — which needs a warp kernel to re-mark it.  A window of
    fewer than 100 events (quirk Q1: no whole batch, nothing is warped) after a resident step launches none: the reference clears num_ev_map
    (model.cpp:85) and finds no inlier and no active pixel; so must the device, instead of the previous window's materialised counts."""
    from emba_amd import EventPacket
    w = small_workload(n_events=20000)
    m = make_legm(w)
    m.set_events(w.events)
    m.upload_map(w.Gx, w.Gy)
    o = oracle_run(oracle_mod, w)
    n_inl, P = m.step(w.traj, w.thres_valid_pixel, w.alpha)
    assert n_inl == o["ep"].size and P == o["ne"]["P"] and P > 0
    ev = w.events
    short = EventPacket(ev.x[:60].copy(), ev.y[:60].copy(), ev.polarity[:60].copy(), ev.t_ns[:60].copy())
    m.set_events(short)
    nem = np.full((w.pano_h, w.pano_w), -7, dtype=np.int32)
    ep = m.evaluateDataError(w.traj, None, None, None, True, nem)
    assert ep.size == 0 and not nem.any()
    ne = m.formNormalEq(None, w.K, nem, w.thres_valid_pixel)
    assert ne["P"] == 0 and ne["active"].size == 0 and not np.any(ne["A11"]) and not np.any(ne["b1"])
    n_inl, P = m.step(w.traj, w.thres_valid_pixel, w.alpha)        # and as a resident step
    assert n_inl == 0 and P == 0
    m.set_events(w.events)                                        # the next window is the first one again
    n_inl, P = m.step(w.traj, w.thres_valid_pixel, w.alpha)
    assert n_inl == o["ep"].size and P == o["ne"]["P"]
    compare_normal_eq(m._finish(w.alpha, False), o["ne"])


def test_call_order_pairs_never_give_a_wrong_number(gpu, oracle_mod):
    """the context carries ~20 phase flags; SURVEY §8b says the boundary is stateful across evaluateDataError -> formNormalEq.
    Table-driven: every ordered PAIR (A, B) of the phase calls, on a fresh context (events + map registered, nothing evaluated) and on a formed
    one, must each either succeed or fail with EMBA_ERR_STATE — and whatever a successful formNormalEq / solve returns must be the oracle's, as
    must the canonical sequence run on the same context afterwards (no pair may leave it in a state that corrupts later valid use)."""
    from emba_amd import EmbaError
    from emba_amd._lib import ERR_STATE
    w = small_workload(n_events=12000, pano_h=128, K=5, sensor=(32, 24), focal=30.0)
    o = oracle_run(oracle_mod, w, dense_A12=True)
    lam = 1e-2
    ox1, ox2 = oracle_mod.solve_normal_eq(o["ne"], lam, True)

    def check_solve(x1, x2):
        assert np.allclose(x1, ox1, rtol=1e-7, atol=1e-9 * np.abs(ox1).max()) and np.allclose(x2, ox2, rtol=1e-7, atol=1e-9 * np.abs(ox2).max())

    ops = {
        "eval_launch": lambda m: m.eval_launch(w.traj),
        "eval_finish": lambda m: m.eval_finish(sync=False),
        "form_active": lambda m: m.form_active(w.thres_valid_pixel),
        "form_accumulate": lambda m: m.form_accumulate(),
        "form_finish": lambda m: dict(ne=m.form_finish(w.alpha, download=True)),
        "solve": lambda m: dict(x=m.solveNormalEq(lam, fix_first_pose=True)),
        "update_map": lambda m: m.updateMap(None, 0.0),
        "map_accept": lambda m: m.acceptMap(),
        "map_reject": lambda m: m.rejectMap(),
        "trial_reject": lambda m: m.rejectTrial(),
        "step": lambda m: m.step(w.traj, w.thres_valid_pixel, w.alpha),
        "get_ep": lambda m: dict(ep=m.get_ep()),         # (must be this window's current evaluation's vector, or EMBA_ERR_STATE — never an earlier one's)
        "set_events": lambda m: m.set_events(w.events),  # ... a new window: whatever was pending belongs to the old one
    }
    n_ok = n_state = 0
    for start in ("fresh", "formed"):
        for a in ops:
            for b in ops:
                m = make_legm(w)
                m.set_events(w.events); m.upload_map(w.Gx, w.Gy)
                if start == "formed":
                    m.step(w.traj, w.thres_valid_pixel, w.alpha)
                trial_map = False           # the device may hold a trial map (zero outside the active set): numbers are then not the oracle's at (Gx, Gy)
                for name in (a, b):
                    try:
                        r = ops[name](m)
                        n_ok += 1
                    except EmbaError as e:
                        assert e.status == ERR_STATE, f"{start}: {a} -> {b}: {name} failed with status {e.status} ({e})"
                        n_state += 1
                        continue
                    if name == "update_map":
                        trial_map = True
                    if name in ("map_accept", "map_reject"):
                        trial_map = trial_map and name == "map_accept"
                    if isinstance(r, dict) and not trial_map:
                        if "ne" in r:
                            compare_normal_eq(r["ne"], o["ne"])
                        elif "ep" in r:
                            assert_close(r["ep"], o["ep"], "ep")
                        else:
                            check_solve(*r["x"])
                # the canonical sequence on the same context afterwards
                nem = np.zeros((w.pano_h, w.pano_w), dtype=np.int32)
                ep = m.evaluateDataError(w.traj, w.Gx, w.Gy, None, True, nem)
                assert np.array_equal(nem, o["num_ev_map"]), f"{start}: {a} -> {b}: count map of the following evaluation"
                assert_close(ep, o["ep"], "ep")
                m.formNormalEq(None, w.K, None, w.thres_valid_pixel)
                compare_normal_eq(m.applyL2Reg(w.alpha), o["ne"])
                check_solve(*m.solveNormalEq(lam, fix_first_pose=True))
                m.close()
    assert n_ok > 100 and n_state > 50, (n_ok, n_state)
