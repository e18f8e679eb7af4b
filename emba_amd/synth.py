"""Deterministic synthetic workload for the hot path (SURVEY.md §8d / BASELINE.md §2).

Everything is a pure function of (seed, sizes): numpy's PCG64 generator, no libc rand, no files.
The same generator feeds the parity tests, __graft_entry__.smoke() and bench.py.
"""
from dataclasses import dataclass

import numpy as np

from .legm import EventPacket, LinearTrajectory

SEED = 20240907


def so3_exp_xyzw(w):
    """Unit quaternion (x,y,z,w) of exp(w) — used only to MAKE inputs (knots), not on the measured path."""
    w = np.asarray(w, dtype=np.float64)
    th = np.linalg.norm(w)
    if th < 1e-12:
        q = np.array([0.5 * w[0], 0.5 * w[1], 0.5 * w[2], 1.0])
    else:
        q = np.concatenate([np.sin(0.5 * th) / th * w, [np.cos(0.5 * th)]])
    return q / np.linalg.norm(q)


def pinhole_bearing_lut(sensor_w, sensor_h, fx, fy, cx, cy):
    """b = ((x-cx)/fx, (y-cy)/fy, 1): what PinholeCameraModel::projectPixelTo3dRay gives for an undistorted
    camera (reference event_pano_warper.cpp:27-41; z = 1, NOT unit-norm)."""
    xs = (np.arange(sensor_w, dtype=np.float64) - cx) / fx
    ys = (np.arange(sensor_h, dtype=np.float64) - cy) / fy
    lut = np.empty((sensor_h, sensor_w, 3), dtype=np.float64)
    lut[..., 0] = xs[None, :]
    lut[..., 1] = ys[:, None]
    lut[..., 2] = 1.0
    return lut.reshape(-1, 3)


def box_blur3(a):
    p = np.pad(a, 1, mode="edge")
    out = np.zeros_like(a)
    for dy in range(3):
        for dx in range(3):
            out += p[dy:dy + a.shape[0], dx:dx + a.shape[1]]
    return out / 9.0


@dataclass
class Workload:
    sensor_w: int
    sensor_h: int
    pano_w: int
    pano_h: int
    lut: np.ndarray
    C_th: float
    Gx: np.ndarray
    Gy: np.ndarray
    traj: LinearTrajectory
    events: EventPacket
    thres_valid_pixel: int = 5
    alpha: float = 5.0
    kind: str = "synthetic"      # "synthetic": i.i.d. uniform events; "scene": simulated from an analytic scene

    @property
    def K(self):
        return self.traj.size()

    def describe(self):
        return (f"{self.kind} N={self.events.size()} sensor={self.sensor_w}x{self.sensor_h} pano={self.pano_h}x{self.pano_w} "
                f"K={self.K} seed={SEED}")


def make_trajectory(K, dt_knots=0.05, t_beg=0.1, yaw_rate=0.5):
    knots = np.stack([so3_exp_xyzw([0.1 * np.sin(2 * np.pi * i / K), yaw_rate * i * dt_knots, 0.05 * np.cos(2 * np.pi * i / K)])
                      for i in range(K)])
    return LinearTrajectory.from_seconds(t_beg, dt_knots, knots)


def make_workload(n_events=1_000_000, pano_h=1024, K=21, sensor=(240, 180), focal=200.0, C_th=0.2, seed=SEED,
                  dt_knots=0.05, t_beg=0.1, yaw_rate=0.5, thres_valid_pixel=5, alpha=5.0):
    """The BASELINE.json configuration by default: 1 M events, 240x180 sensor, 1024x2048 panorama, K=21 (T = 1 s)."""
    rng = np.random.default_rng(seed)
    sw, sh = sensor
    pano_w = 2 * pano_h
    lut = pinhole_bearing_lut(sw, sh, focal, focal, sw / 2.0, sh / 2.0)
    Gx = box_blur3(rng.normal(0.0, 0.1 * C_th, size=(pano_h, pano_w)))
    Gy = box_blur3(rng.normal(0.0, 0.1 * C_th, size=(pano_h, pano_w)))
    traj = make_trajectory(K, dt_knots, t_beg, yaw_rate)
    T_ns = traj.dt_ns * (K - 1)
    n = int(n_events)
    # strictly increasing, uniformly spaced timestamps inside [t0, t0 + T)
    t_ns = traj.t0_ns + (np.arange(n, dtype=np.int64) * T_ns) // max(n, 1)
    x = rng.integers(0, sw, size=n, dtype=np.uint16)
    y = rng.integers(0, sh, size=n, dtype=np.uint16)
    pol = rng.integers(0, 2, size=n, dtype=np.uint8)
    return Workload(sw, sh, pano_w, pano_h, lut, C_th, Gx, Gy, traj, EventPacket(x, y, pol, t_ns), thres_valid_pixel, alpha)


# ---- a physically consistent scene: events generated FROM a panorama and a rotation, for end-to-end checks of the LM loop ------
class SinusoidScene:
    """Log-intensity panorama L(px, py) = sum_k a_k sin(2 pi m_k px / W + phi_k) cos(pi n_k py / H + psi_k): periodic in px,
    analytic gradient, so the gradient map handed to the solver is the exact derivative the event generator integrates."""

    def __init__(self, pano_h, n_terms=12, max_freq=(14, 7), amp=1.0, seed=3):
        rng = np.random.default_rng(seed)
        self.H, self.W = int(pano_h), 2 * int(pano_h)
        self.m = rng.integers(1, max_freq[0] + 1, size=n_terms).astype(np.float64)
        self.n = rng.integers(0, max_freq[1] + 1, size=n_terms).astype(np.float64)
        self.a = amp * rng.uniform(0.3, 1.0, size=n_terms) / np.sqrt(n_terms)
        self.phi = rng.uniform(0, 2 * np.pi, size=n_terms)
        self.psi = rng.uniform(0, 2 * np.pi, size=n_terms)

    def value(self, px, py):
        ax = 2 * np.pi * self.m * px[..., None] / self.W + self.phi
        ay = np.pi * self.n * py[..., None] / self.H + self.psi
        return (self.a * np.sin(ax) * np.cos(ay)).sum(-1)

    def gradient_map(self):
        """(Gx, Gy) at integer pixel coordinates (the hot path samples G at round(p), model.cpp:209-217)."""
        py, px = np.meshgrid(np.arange(self.H, dtype=np.float64), np.arange(self.W, dtype=np.float64), indexing="ij")
        ax = 2 * np.pi * self.m * px[..., None] / self.W + self.phi
        ay = np.pi * self.n * py[..., None] / self.H + self.psi
        Gx = (self.a * (2 * np.pi * self.m / self.W) * np.cos(ax) * np.cos(ay)).sum(-1)
        Gy = (-self.a * (np.pi * self.n / self.H) * np.sin(ax) * np.sin(ay)).sum(-1)
        return np.ascontiguousarray(Gx), np.ascontiguousarray(Gy)


def _quat_to_R(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def project_equirect(pano_w, pano_h, P):
    """Equirectangular projection of bearing vectors [n,3] (reference include/utils/equirectangular_camera.h:25-40)."""
    fx = pano_w / (2 * np.pi); fy = pano_h / np.pi
    rho = np.linalg.norm(P, axis=-1)
    return pano_w / 2.0 + np.arctan2(P[..., 0], P[..., 2]) * fx, pano_h / 2.0 + np.arcsin(P[..., 1] / rho) * fy


def simulate_events(scene, traj, lut, sensor, C_th, n_steps=4000):
    """Ideal event camera rotating along `traj` in front of `scene`: each sensor pixel fires (polarity = sign) whenever its
    log intensity has changed by C_th since its last event; timestamps by linear interpolation inside a time step."""
    from . import so3
    sw, sh = sensor
    knots = traj.knots_xyzw
    K = knots.shape[0]
    t_end = traj.t0_ns + traj.dt_ns * (K - 1)
    ts = np.linspace(traj.t0_ns, t_end - 1, n_steps + 1)
    L_prev = L_ref = None
    ex, ey, ep, et = [], [], [], []
    pix = np.arange(sw * sh)
    for s, t in enumerate(ts):
        i = min(int((t - traj.t0_ns) // traj.dt_ns), K - 2)
        u = (t - traj.t0_ns - i * traj.dt_ns) / traj.dt_ns
        q = so3.mul(knots[i], so3.exp(u * so3.log(so3.mul(so3.inverse(knots[i]), knots[i + 1]))))
        px, py = project_equirect(scene.W, scene.H, lut @ _quat_to_R(q).T)
        L = scene.value(px, py)
        if s == 0:
            L_prev, L_ref = L, L.copy()
            continue
        while True:
            d = L - L_ref
            fire = np.abs(d) >= C_th
            if not fire.any():
                break
            sgn = np.sign(d[fire])
            L_ref[fire] += sgn * C_th
            frac = np.clip((L_ref[fire] - L_prev[fire]) / (L[fire] - L_prev[fire]), 0.0, 1.0)
            ex.append(pix[fire] % sw); ey.append(pix[fire] // sw); ep.append((sgn > 0).astype(np.uint8))
            et.append((ts[s - 1] + frac * (t - ts[s - 1])).astype(np.int64))
        L_prev = L
    x = np.concatenate(ex).astype(np.uint16); y = np.concatenate(ey).astype(np.uint16)
    p = np.concatenate(ep); t = np.concatenate(et)
    order = np.argsort(t, kind="stable")
    return EventPacket(x[order], y[order], p[order], t[order])


def make_scene_workload(pano_h=256, K=6, sensor=(64, 48), focal=60.0, C_th=0.2, dt_knots=0.05, t_beg=0.1, yaw_rate=0.5, amp=8.0,
                        n_steps=4000, seed=3, thres_valid_pixel=5, alpha=5.0, n_terms=12, max_freq=(14, 7)):
    """Ground-truth trajectory + analytic scene + the events that motion produces; Gx, Gy are the TRUE gradient map."""
    sw, sh = sensor
    lut = pinhole_bearing_lut(sw, sh, focal, focal, sw / 2.0, sh / 2.0)
    scene = SinusoidScene(pano_h, n_terms=n_terms, max_freq=max_freq, amp=amp, seed=seed)
    Gx, Gy = scene.gradient_map()
    traj = make_trajectory(K, dt_knots, t_beg, yaw_rate)
    ev = simulate_events(scene, traj, lut, sensor, C_th, n_steps)
    return Workload(sw, sh, 2 * pano_h, pano_h, lut, C_th, Gx, Gy, traj, ev, thres_valid_pixel, alpha, "scene")


def make_scene_stream(n_target, pano_h=1024, K=21, sensor=(240, 180), focal=200.0, C_th=0.2, n_steps=400, seed=3, **kw):
    """A scene-driven event stream of about n_target events (bench.py --data scene): events cluster where the scene has contrast,
    every sensor pixel fires at its own rate and polarities follow the sign of the brightness change — unlike the i.i.d. uniform
    workload of SURVEY §8d.  The scene amplitude is calibrated on a coarse run so that the count lands near the target."""
    sw, sh = sensor
    lut = pinhole_bearing_lut(sw, sh, focal, focal, sw / 2.0, sh / 2.0)
    traj = make_trajectory(K)
    probe = SinusoidScene(pano_h, n_terms=8, max_freq=(40, 20), amp=1.0, seed=seed)
    c = simulate_events(probe, traj, lut, sensor, C_th, max(n_steps // 8, 20)).size()
    amp = float(n_target) / max(c, 1)
    w = make_scene_workload(pano_h=pano_h, K=K, sensor=sensor, focal=focal, C_th=C_th, amp=amp, n_steps=n_steps, seed=seed,
                            n_terms=8, max_freq=(40, 20), **kw)
    return w
