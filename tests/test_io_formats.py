"""the ROS-free data formats and trajectory plumbing either side of the hot path (emba_amd/io.py, so3.py) —
host logic, checked against the pinned oracle where the same operation exists there."""
import numpy as np
import pytest

from emba_amd import io as eio
from emba_amd import so3
from emba_amd.legm import EventPacket, LinearTrajectory
from emba_amd.synth import make_trajectory


def test_so3_helpers_match_pinned_oracle(oracle_mod):
    rng = np.random.default_rng(0)
    for scale in (1e-12, 1e-6, 0.3, 2.5):
        for _ in range(20):
            w = rng.normal(size=3) * scale
            q = so3.exp(w)
            assert np.allclose(q, oracle_mod.so3_exp(w), rtol=0, atol=1e-15)
            assert np.allclose(so3.log(q), oracle_mod.so3_log(q), rtol=1e-12, atol=1e-18)
            if np.linalg.norm(w) < np.pi:                       # beyond pi the log returns the equivalent shorter rotation, as Sophus does
                assert np.allclose(so3.log(q), w, rtol=1e-9, atol=1e-18)
    a, b = so3.exp([0.1, -0.2, 0.3]), so3.exp([-0.4, 0.1, 0.2])
    assert np.allclose(so3.mul(so3.mul(a, b), so3.inverse(b)), a, atol=1e-15)


def test_map_files_round_trip(tmp_path):
    rng = np.random.default_rng(1)
    Gx, Gy = rng.normal(size=(32, 64)), rng.normal(size=(32, 64))
    eio.save_map(tmp_path / "map", Gx, Gy)
    assert (tmp_path / "map" / "Gx.bin").stat().st_size == 32 * 64 * 8       # raw doubles, no header (emba.cpp:541-563)
    gx, gy = eio.load_map(tmp_path / "map")
    assert gx.shape == (32, 64) and np.array_equal(gx, Gx) and np.array_equal(gy, Gy)
    np.zeros(2 * 64 * 64).tofile(tmp_path / "map" / "Gy.bin")
    with pytest.raises(ValueError):
        eio.load_map(tmp_path / "map")                                         # sizes differ
    np.zeros(100).tofile(tmp_path / "map" / "Gx.bin"); np.zeros(100).tofile(tmp_path / "map" / "Gy.bin")
    with pytest.raises(ValueError):
        eio.load_map(tmp_path / "map")                                         # not H x 2H


def test_pose_file_and_interpolation(tmp_path, oracle_mod):
    traj = make_trajectory(5, dt_knots=0.05, t_beg=0.1)
    path = tmp_path / "poses.txt"
    with open(path, "w") as f:
        f.write("# timestamp tx ty tz qx qy qz qw\n")
        for i, q in enumerate(traj.knots_xyzw):
            f.write(f"{0.1 + 0.05 * i:.9f} 1 2 3 {2 * q[0]:.17g} {2 * q[1]:.17g} {2 * q[2]:.17g} {2 * q[3]:.17g}\n")    # unnormalised on purpose
        f.write("garbage line\n\n")
    t, qs = eio.load_poses(path, time_offset=0.5)
    assert t.shape == (5,) and np.allclose(t, 0.6 + 0.05 * np.arange(5)) and np.allclose(qs, traj.knots_xyzw, atol=1e-15)
    t -= 0.5
    # between two samples the interpolation IS the linear SO(3) spline through them (pose_manager.cpp:100-107 vs so3_spline.h:213-243)
    for tq in (0.1, 0.1125, 0.16, 0.2999):
        q, *_ = oracle_mod.spline_eval(traj.knots_xyzw, traj.t0_ns, traj.dt_ns, int(round(tq * 1e9)))
        p = eio.pose_at(t, qs, tq)
        assert min(np.abs(p - q).max(), np.abs(p + q).max()) < 1e-9
    assert np.array_equal(eio.pose_at(t, qs, 0.0), qs[0]) and np.array_equal(eio.pose_at(t, qs, 9.0), qs[-1])     # clamped


def test_trajectory_file(tmp_path):
    traj = make_trajectory(4, dt_knots=0.05, t_beg=0.1)
    eio.write_trajectory(tmp_path / "traj.txt", traj, time_offset=0.02)
    lines = open(tmp_path / "traj.txt").read().splitlines()
    assert len(lines) == 4 and all(len(l.split()) == 8 for l in lines)
    assert lines[0].split()[:4] == ["0.08", "0", "0", "0"]
    t, qs = eio.load_poses(tmp_path / "traj.txt", time_offset=0.02)
    assert np.allclose(t, [0.1, 0.15, 0.2, 0.25]) and np.allclose(qs, traj.knots_xyzw, atol=1e-5)      # 6 significant digits


def test_fit_ctrl_poses_recovers_a_spline(oracle_mod):
    # rotation about one axis: the tangent-space fit is exact
    knots = np.stack([so3.exp([0.0, 0.3 * i * i * 0.05, 0.0]) for i in range(6)])
    traj = LinearTrajectory.from_seconds(0.1, 0.05, knots)
    ts = np.linspace(0.1, 0.35 - 1e-6, 200)
    qs = np.stack([oracle_mod.spline_eval(knots, traj.t0_ns, traj.dt_ns, int(t * 1e9))[0] for t in ts])
    fit = eio.fit_ctrl_poses(ts, qs, 0.1, 0.05, 6)
    assert np.abs(fit - knots).max() < 1e-6
    # general small rotations: close (the spline is geodesic between knots, the fit linear in one tangent space)
    traj = make_trajectory(6, dt_knots=0.05, t_beg=0.1)
    qs = np.stack([oracle_mod.spline_eval(traj.knots_xyzw, traj.t0_ns, traj.dt_ns, int(t * 1e9))[0] for t in ts])
    fit = eio.fit_ctrl_poses(ts, qs, 0.1, 0.05, 6)
    ang = [np.linalg.norm(so3.log(so3.mul(so3.inverse(a), b))) for a, b in zip(fit, traj.knots_xyzw)]
    assert max(ang) < 2e-3
    with pytest.raises(ValueError):
        eio.fit_ctrl_poses(ts[:3], qs[:3], 0.1, 0.05, 6)
    long = eio.generate_ctrl_poses_long(ts, qs, 0.1, 0.35, 0.05, 0.1)
    assert long.shape == (2 * 2 + 1, 4)                                        # 2 sub-windows of 3 control poses sharing one


def test_incremental_update_is_a_left_perturbation(oracle_mod):
    traj = make_trajectory(4)
    x1 = np.random.default_rng(2).normal(size=12) * 0.05
    new = eio.incremental_update(traj, x1, fix_first_pose=True)
    assert np.array_equal(new.knots_xyzw[0], traj.knots_xyzw[0])
    for i in range(1, 4):
        d = so3.mul(new.knots_xyzw[i], so3.inverse(traj.knots_xyzw[i]))        # = exp(x1_i)
        assert np.allclose(oracle_mod.so3_log(d), x1[3 * i:3 * i + 3], atol=1e-12)
    assert not np.array_equal(eio.incremental_update(traj, x1, False).knots_xyzw[0], traj.knots_xyzw[0])


def test_event_file_round_trip(tmp_path):
    rng = np.random.default_rng(3)
    n = 1000
    ev = EventPacket(rng.integers(0, 240, n).astype(np.uint16), rng.integers(0, 180, n).astype(np.uint16),
                     rng.integers(0, 2, n).astype(np.uint8), rng.permutation(n).astype(np.int64) * 1000)
    eio.save_events(tmp_path / "ev.npz", ev)
    back = eio.load_events(tmp_path / "ev.npz")
    assert (np.diff(back.t_ns) >= 0).all() and back.size() == n
    order = np.argsort(ev.t_ns, kind="stable")
    assert np.array_equal(back.x, ev.x[order]) and np.array_equal(back.polarity, ev.polarity[order])
    win = eio.load_events(tmp_path / "ev.npz", t_min_ns=100_000, t_max_ns=200_000)
    assert win.size() == 101 and win.t_ns[0] == 100_000 and win.t_ns[-1] == 200_000


def test_bearing_lut_from_calibration():
    from emba_amd.synth import pinhole_bearing_lut
    K = np.array([[200.0, 0, 120.0], [0, 200.0, 90.0], [0, 0, 1]])
    lut = eio.bearing_lut_from_calibration(K, None, 240, 180)
    ref = pinhole_bearing_lut(240, 180, 200.0, 200.0, 120.0, 90.0)
    assert lut.shape == (240 * 180, 3)
    assert np.allclose(lut / np.linalg.norm(lut, axis=1, keepdims=True), ref / np.linalg.norm(ref, axis=1, keepdims=True), atol=1e-15)
    # with distortion: re-distorting the undistorted point returns the pixel
    D = np.array([-0.3, 0.1, 1e-3, -2e-3, 0.0])
    lut = eio.bearing_lut_from_calibration(K, D, 240, 180, iters=50)
    x, y = lut[:, 0], lut[:, 1]
    r2 = x * x + y * y
    rad = 1 + D[0] * r2 + D[1] * r2 ** 2 + D[4] * r2 ** 3
    xd = x * rad + 2 * D[2] * x * y + D[3] * (r2 + 2 * x * x)
    yd = y * rad + D[2] * (r2 + 2 * y * y) + 2 * D[3] * x * y
    v, u = np.meshgrid(np.arange(180.0), np.arange(240.0), indexing="ij")
    assert np.abs(xd * 200 + 120 - u.ravel()).max() < 1e-6 and np.abs(yd * 200 + 90 - v.ravel()).max() < 1e-6


def test_normalize_robust_and_pgm(tmp_path):
    rng = np.random.default_rng(4)
    img = rng.normal(size=(40, 50)); img[0, 0] = 1e6; img[1, 1] = -1e6          # outliers must not set the range
    u8 = eio.normalize_robust(img, 1.0)
    assert u8.dtype == np.uint8 and u8.min() == 0 and u8.max() == 255 and 100 < np.median(u8) < 160
    srt = np.sort(img, axis=None)
    lo, hi = srt[int(0.005 * 2000)], srt[int(0.995 * 2000)]
    k = (5, 7)
    assert u8[k] == np.clip(np.rint(255.0 / (hi - lo) * (img[k] - lo)), 0, 255)
    assert (eio.normalize_robust(np.full((4, 4), 3.0)) == 0).all()            # rmax == rmin: scale 1
    eio.save_pgm(tmp_path / "a.pgm", u8)
    raw = open(tmp_path / "a.pgm", "rb").read()
    assert raw.startswith(b"P5\n50 40\n255\n") and len(raw) == len(b"P5\n50 40\n255\n") + 2000
