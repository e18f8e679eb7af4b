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

    @property
    def K(self):
        return self.traj.size()

    def describe(self):
        return (f"synthetic N={self.events.size()} sensor={self.sensor_w}x{self.sensor_h} pano={self.pano_h}x{self.pano_w} "
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
