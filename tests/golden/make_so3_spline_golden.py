"""Generates tests/golden/so3_spline_n2.npz from the REFERENCE's own code (oracle/_ref/libref_basalt.so =
unmodified basalt::So3Spline<2>::evaluate, so3_spline.h:218-274, with vendored Sophus + Eigen, compiled by
oracle/Makefile from /root/reference).  Run in the authoring container only:

    make -C oracle && python tests/golden/make_so3_spline_golden.py

The fixture holds inputs (knots, t0, dt, t) and the reference's outputs (q, R, start_idx, J 3x6); it is data, not
source.  Knots are pre-normalised to a fixed point of Sophus' SO3(Quaternion) constructor so that the extra
normalisation done by knotsPushBack (SO3 ctor, so3.hpp:481-487) is the identity and both sides see identical knots.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from oracle import oracle as O  # noqa: E402


def norm_sophus(q):
    n2 = (q[0] * q[0] + q[2] * q[2]) + (q[1] * q[1] + q[3] * q[3])
    return q / np.sqrt(n2)


def stable(q):
    for _ in range(20):
        q2 = norm_sophus(q)
        if (q2 == q).all():
            return q
        q = q2
    return None


def main():
    assert O.ref_available(), "oracle/_ref not built (needs /root/reference)"
    rng = np.random.default_rng(20241022)
    T0, DT = 100_000_000, 50_000_000
    cases = []

    def add(knots, t, tag):
        ks = [stable(np.asarray(k, dtype=np.float64)) for k in knots]
        if any(k is None for k in ks):
            return False
        ks = np.array(ks)
        r = O.spline_eval(ks, T0, DT, t, use_ref=True)
        assert r is not None
        q, R, s, J = r
        cases.append(dict(knots=ks, t=t, q=q, R=R, s=s, J=J, tag=tag))
        return True

    K = 6
    n = 0
    while n < 160:  # general
        w = rng.normal(size=3) * 0.4
        knots = []
        for i in range(K):
            w = w + rng.normal(size=3) * rng.choice([0.02, 0.1, 0.5])
            knots.append(O.so3_exp(w, use_ref=True))
        n += add(knots, T0 + int(rng.integers(0, DT * (K - 1))), "general")
    n = 0
    while n < 40:  # tiny relative rotation -> Taylor branches of Jl / Jl^-1 and of exp
        w = rng.normal(size=3)
        knots = [O.so3_exp(w + rng.normal(size=3) * s, use_ref=True) for s in (0, 1e-9, 1e-7, 1e-6, 3e-6, 1e-5)]
        n += add(knots, T0 + int(rng.integers(0, DT * (K - 1))), "tiny")
    n = 0
    while n < 10:  # identical knots -> log small-angle branch
        q = O.so3_exp(rng.normal(size=3), use_ref=True)
        n += add([q] * K, T0 + int(rng.integers(0, DT * (K - 1))), "identical")
    n = 0
    while n < 20:  # u == 0 exactly (query on a knot)
        w = rng.normal(size=3) * 0.3
        knots = [O.so3_exp(w + rng.normal(size=3) * 0.1, use_ref=True) for _ in range(K)]
        n += add(knots, T0 + DT * int(rng.integers(0, K - 1)), "on_knot")
    n = 0
    while n < 30:  # relative rotation close to pi -> near-pi branch of Jl^-1, large-angle log
        axis = rng.normal(size=3); axis /= np.linalg.norm(axis)
        ang = np.pi - rng.choice([1e-7, 1e-6, 1e-4, 1e-2, 0.2])
        base = O.so3_exp(rng.normal(size=3) * 0.2, use_ref=True)
        knots = [base, base, base, base, base, base]
        # knot1 = knot0 * exp(ang*axis): compose with the reference's product by way of exp/log-free quaternion algebra
        e = O.so3_exp(axis * ang, use_ref=True)
        ax, ay, az, aw = base; bx, by, bz, bw = e
        prod = np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by + ay * bw + az * bx - ax * bz,
                         aw * bz + az * bw + ax * by - ay * bx, aw * bw - ax * bx - ay * by - az * bz])
        knots[1] = prod
        n += add(knots, T0 + int(rng.integers(0, DT)), "near_pi")

    out = os.path.join(os.path.dirname(__file__), "so3_spline_n2.npz")
    np.savez_compressed(out, t0_ns=T0, dt_ns=DT, knots=np.array([c["knots"] for c in cases]),
                        t=np.array([c["t"] for c in cases], dtype=np.int64), q=np.array([c["q"] for c in cases]),
                        R=np.array([c["R"] for c in cases]), s=np.array([c["s"] for c in cases], dtype=np.int32),
                        J=np.array([c["J"] for c in cases]), tag=np.array([c["tag"] for c in cases]))
    print("wrote", out, len(cases), "cases", os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
