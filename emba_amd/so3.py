"""Small SO(3) helpers on unit quaternions (x, y, z, w) for the HOST-side, per-control-pose steps of the callers
(trajectory fitting and update: O(K) work the reference also does on the host).  Formulas follow Sophus
(so3.hpp:247-290 log, :583-619 exp, :324-339 product); nothing here runs per event."""
import numpy as np

EPS = 1e-10


def normalize(q):
    q = np.asarray(q, dtype=np.float64)
    return q / np.linalg.norm(q)


def mul(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return normalize(np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by + ay * bw + az * bx - ax * bz,
                               aw * bz + az * bw + ax * by - ay * bx, aw * bw - ax * bx - ay * by - az * bz]))


def inverse(q):
    return normalize(np.array([-q[0], -q[1], -q[2], q[3]]))


def exp(w):
    w = np.asarray(w, dtype=np.float64)
    th2 = float(w @ w)
    if th2 < EPS * EPS:
        imag = 0.5 - th2 / 48.0 + th2 * th2 / 3840.0
        real = 1.0 - th2 / 8.0 + th2 * th2 / 384.0
    else:
        th = np.sqrt(th2)
        imag = np.sin(0.5 * th) / th
        real = np.cos(0.5 * th)
    return np.array([imag * w[0], imag * w[1], imag * w[2], real])


def log(q):
    q = np.asarray(q, dtype=np.float64)
    n2 = float(q[:3] @ q[:3]); w = q[3]
    if n2 < EPS * EPS:
        f = 2.0 / w - (2.0 / 3.0) * n2 / (w * w * w)
    else:
        n = np.sqrt(n2)
        f = (np.pi / n if w > 0 else -np.pi / n) if abs(w) < EPS else 2.0 * np.arctan(n / w) / n
    return f * q[:3]
