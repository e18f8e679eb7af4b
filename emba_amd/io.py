"""Data formats and trajectory plumbing on either side of the hot path, without ROS.  Mirrors:

  load_map / save_map            EMBA::loadMap                       src/emba/emba.cpp:535-578   (Gx.bin / Gy.bin, raw f64 row-major,
                                                                                                  H = sqrt(n/2), W = 2H)
  load_poses                     PoseManager::loadPoses              src/utils/pose_manager.cpp:41-80  ("ts tx ty tz qx qy qz qw" per line)
  pose_at                        PoseManager::getPoseAt              :82-108                      (geodesic interpolation)
  write_trajectory               LinearTrajectory::write             src/utils/trajectory.cpp:98-114
  fit_ctrl_poses                 LinearTrajectory::fitCtrlPoses      :149-229                     (tangent-space least squares)
  generate_ctrl_poses_long       LinearTrajectory::generateCtrlPosesLong  :258-294
  incremental_update             LinearTrajectory::incrementalUpdate :296-304                     (left perturbation exp(x)*knot)
  bearing_lut_from_calibration   EventWarper::precomputeBearingVectors src/utils/event_pano_warper.cpp:27-41 (image_geometry rectifyPoint +
                                                                      projectPixelTo3dRay for a monocular plumb_bob camera)
  normalize_robust / save_pgm    image_util::normalizeRobust          src/utils/image_utils.cpp:13-38   (8-bit display images of maps and of the
                                                                      Poisson-reconstructed panorama, solver.cpp:417-425; PGM instead of PNG)
  save_events / load_events      a flat .npz replacing the rosbag of src/utils/rosbag_loading.cpp (x, y u16; polarity u8; t_ns i64, sorted)

Host-side, O(K) or file-sized work; the per-event path is emba_amd.LEGM.
"""
import os

import numpy as np

from . import so3
from .legm import EventPacket, LinearTrajectory


# ---- panoramic gradient map ------------------------------------------------------------------------------------------
def load_map(map_dir):
    gx = np.fromfile(os.path.join(map_dir, "Gx.bin"), dtype="<f8")
    gy = np.fromfile(os.path.join(map_dir, "Gy.bin"), dtype="<f8")
    if gx.size != gy.size:
        raise ValueError("Gx.bin and Gy.bin differ in size")                    # CHECK_EQ, emba.cpp:566
    H = int(np.sqrt(gx.size / 2))                                               # emba.cpp:552
    W = 2 * H
    if H * W != gx.size:
        raise ValueError(f"{gx.size} doubles is not an H x 2H panorama")
    return gx.reshape(H, W).copy(), gy.reshape(H, W).copy()


def save_map(map_dir, Gx, Gy):
    os.makedirs(map_dir, exist_ok=True)
    np.ascontiguousarray(Gx, dtype="<f8").tofile(os.path.join(map_dir, "Gx.bin"))
    np.ascontiguousarray(Gy, dtype="<f8").tofile(os.path.join(map_dir, "Gy.bin"))


# ---- poses --------------------------------------------------------------------------------------------------------------
def load_poses(path, time_offset=0.0):
    """Returns (t [n] seconds ascending, q [n,4] xyzw unit); lines that do not parse as 8 numbers are skipped like the reference's `if (ss >> ...)`."""
    ts, qs = [], []
    with open(path) as f:
        for line in f:
            parts = line.split()
            if len(parts) < 8:
                continue
            try:
                v = [float(x) for x in parts[:8]]
            except ValueError:
                continue
            ts.append(v[0] + time_offset)
            qs.append(so3.normalize(v[4:8]))                                   # Sophus::SO3d(q) normalises
    order = np.argsort(np.array(ts), kind="stable")                            # std::map<ros::Time, SO3d>
    return np.array(ts)[order], np.array(qs).reshape(-1, 4)[order]


def pose_at(t, qs, t_query):
    """PoseManager::getPoseAt: clamp outside, R1 * exp(s * log(R1^-1 R2)) inside."""
    i2 = int(np.searchsorted(t, t_query, side="right"))                        # upper_bound
    if i2 == 0:
        return qs[0]
    if i2 == len(t):
        return qs[-1]
    q1, q2 = qs[i2 - 1], qs[i2]
    s = (t_query - t[i2 - 1]) / (t[i2] - t[i2 - 1])
    return so3.mul(q1, so3.exp(s * so3.log(so3.mul(so3.inverse(q1), q2))))


def write_trajectory(path, traj, time_offset=0.0):
    """`t 0 0 0 qx qy qz qw` per control pose; numbers as an std::ofstream prints doubles by default (%g, 6 significant digits)."""
    t_beg = traj.t0_ns * 1e-9
    dt = traj.dt_ns * 1e-9
    with open(path, "w") as f:
        for i, q in enumerate(traj.knots_xyzw):
            f.write("%g 0 0 0 %g %g %g %g\n" % (t_beg + i * dt - time_offset, q[0], q[1], q[2], q[3]))


# ---- control poses --------------------------------------------------------------------------------------------------------
def fit_ctrl_poses(t, qs, t_beg, dt_knots, num_cps):
    """LinearTrajectory::fitCtrlPoses: lift to the tangent space at the first pose, solve N P = D for the control points of the
    uniform linear spline (basis [1-u, u]), retract."""
    if len(t) < num_cps:
        raise ValueError("fewer poses than control poses")                     # CHECK_GE
    offset = qs[0]
    off_inv = so3.inverse(offset)
    N = np.zeros((len(t), num_cps))
    D = np.zeros((len(t), 3))
    for k, (tk, qk) in enumerate(zip(t, qs)):
        ti = int(np.floor((tk - t_beg) / dt_knots))
        u = (tk - (ti * dt_knots + t_beg)) / dt_knots
        N[k, ti] = 1.0 - u                                                     # U * M2 with M2 = [[1,0],[-1,1]]
        if ti + 1 < num_cps:
            N[k, ti + 1] = u
        D[k] = so3.log(so3.mul(off_inv, qk))
    P = np.linalg.lstsq(N, D, rcond=None)[0]                                   # fullPivHouseholderQr().solve
    return np.array([so3.mul(offset, so3.exp(P[i])) for i in range(num_cps)])


def generate_ctrl_poses_long(t, qs, t_beg, t_end, dt_knots, sub_interval_length):
    n_sub = int(np.floor((t_end - t_beg) / sub_interval_length + 1e-6))
    out = []
    for i in range(n_sub):
        a = t_beg + sub_interval_length * i
        b = a + sub_interval_length
        sel = (t > a) & (t < b)                                                # upper_bound(a) .. lower_bound(b)
        num_cps = int(round((b - a) / dt_knots)) + 1
        cps = fit_ctrl_poses(t[sel], qs[sel], a, dt_knots, num_cps)
        out.extend(cps[1:] if i else cps)
    return np.array(out)


def incremental_update(traj, x1, fix_first_pose):
    """Model::updateTraj + LinearTrajectory::incrementalUpdate (model.cpp:22-53, trajectory.cpp:296-304): knot_i <- exp(x1_i) * knot_i.
    x1 has 3K entries (zeros for a fixed first pose, as emba_solve_normal_eq returns it).  All control poses at once (numpy; the same
    Sophus formulas as so3.exp / so3.mul: so3.hpp:583-619, 324-339) — the per-pose Python loop was 1.1 ms of a 8.4-ms LM iteration at K = 201."""
    knots = traj.knots_xyzw.copy()
    s = 1 if fix_first_pose else 0
    w = np.asarray(x1, dtype=np.float64).reshape(-1, 3)[s:len(knots)]
    th2 = (w * w).sum(axis=1)
    small = th2 < so3.EPS * so3.EPS
    th = np.sqrt(np.where(small, 1.0, th2))
    imag = np.where(small, 0.5 - th2 / 48.0 + th2 * th2 / 3840.0, np.sin(0.5 * th) / th)
    real = np.where(small, 1.0 - th2 / 8.0 + th2 * th2 / 384.0, np.cos(0.5 * th))
    ax, ay, az, aw = imag * w[:, 0], imag * w[:, 1], imag * w[:, 2], real
    bx, by, bz, bw = knots[s:, 0], knots[s:, 1], knots[s:, 2], knots[s:, 3]
    q = np.stack([aw * bx + ax * bw + ay * bz - az * by, aw * by + ay * bw + az * bx - ax * bz,
                  aw * bz + az * bw + ax * by - ay * bx, aw * bw - ax * bx - ay * by - az * bz], axis=1)
    knots[s:] = q / np.linalg.norm(q, axis=1, keepdims=True)
    return LinearTrajectory(knots, traj.t0_ns, traj.dt_ns)


# ---- camera ---------------------------------------------------------------------------------------------------------------
def bearing_lut_from_calibration(K, D, width, height, iters=5):
    """Bearing vector (x', y', 1) of every sensor pixel, row-major [height*width, 3], for a monocular plumb_bob camera (R = I,
    P = [K | 0]): rectifyPoint undistorts the pixel (cv::undistortPoints: fixed-point iteration on the radial-tangential model,
    5 iterations) and re-projects with K; projectPixelTo3dRay maps that back through K, so the ray is the undistorted normalised
    point.  image_geometry / OpenCV are not part of the reference tree: published algorithm, parity unpinned."""
    K = np.asarray(K, dtype=np.float64).reshape(3, 3)
    D = np.zeros(5) if D is None else np.concatenate([np.asarray(D, dtype=np.float64).ravel(), np.zeros(5)])[:5]
    k1, k2, p1, p2, k3 = D
    v, u = np.meshgrid(np.arange(height, dtype=np.float64), np.arange(width, dtype=np.float64), indexing="ij")
    x0 = (u - K[0, 2]) / K[0, 0]
    y0 = (v - K[1, 2]) / K[1, 1]
    x, y = x0.copy(), y0.copy()
    for _ in range(iters):
        r2 = x * x + y * y
        icdist = 1.0 / (1.0 + ((k3 * r2 + k2) * r2 + k1) * r2)
        dx = 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
        dy = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
        x = (x0 - dx) * icdist
        y = (y0 - dy) * icdist
    return np.ascontiguousarray(np.stack([x, y, np.ones_like(x)], axis=-1).reshape(-1, 3))


# ---- display images ---------------------------------------------------------------------------------------------------------
def normalize_robust(img, percentage_pixels_to_discard=0.1):
    """image_util::normalizeRobust: scale [robust min, robust max] (order statistics after discarding the given percentage of
    pixels, float32 index arithmetic like the reference) to [0, 255]; cv::Mat::convertTo(CV_8UC1) = round half to even, saturate."""
    a = np.asarray(img, dtype=np.float64)
    srt = np.sort(a, axis=None)
    n = a.size
    i_min = int(np.float32(np.float32(0.5) * np.float32(percentage_pixels_to_discard) / np.float32(100.0)) * np.float32(n))
    i_max = int(np.float32(np.float32(1.0) - np.float32(0.5) * np.float32(percentage_pixels_to_discard) / np.float32(100.0)) * np.float32(n))
    rmin, rmax = srt[i_min], srt[min(i_max, n - 1)]
    scale = 255.0 / (rmax - rmin) if rmax != rmin else 1.0
    return np.clip(np.rint(scale * (a - rmin)), 0, 255).astype(np.uint8)


def save_pgm(path, img_u8):
    img_u8 = np.ascontiguousarray(img_u8, dtype=np.uint8)
    with open(path, "wb") as f:
        f.write(b"P5\n%d %d\n255\n" % (img_u8.shape[1], img_u8.shape[0]))
        f.write(img_u8.tobytes())


# ---- events ---------------------------------------------------------------------------------------------------------------
def save_events(path, events):
    np.savez_compressed(path, x=np.asarray(events.x, np.uint16), y=np.asarray(events.y, np.uint16),
                        polarity=np.asarray(events.polarity, np.uint8), t_ns=np.asarray(events.t_ns, np.int64))


def load_events(path, t_min_ns=None, t_max_ns=None):
    """Events in [t_min, t_max], sorted by timestamp (stable) like parse_rosbag's std::sort (rosbag_loading.cpp:61-65)."""
    d = np.load(path)
    x, y, p, t = d["x"], d["y"], d["polarity"], d["t_ns"]
    order = np.argsort(t, kind="stable")
    x, y, p, t = x[order], y[order], p[order], t[order]
    sel = np.ones(t.size, bool)
    if t_min_ns is not None:
        sel &= t >= t_min_ns
    if t_max_ns is not None:
        sel &= t <= t_max_ns
    return EventPacket(x[sel], y[sel], p[sel], t[sel])
