"""Generates tests/golden/eigen_solvers.npz from the REFERENCE's own vendored Eigen (oracle/_ref/libref_eigen.so = oracle/ref_eigen.cpp
compiled against /root/reference/thirdparty/basalt-headers/thirdparty/eigen by oracle/Makefile): the three library calls of the
reference's solvers on matrices the oracle produces —

    S.ldlt().solve(rhs)                                  src/emba/model.cpp:789
    A22m_i.inverse()   (Eigen::Matrix2d)                 src/emba/model.cpp:750
    ConjugateGradient<SpMat, Lower|Upper>(100, 1e-6)     src/emba/model.cpp:828-836

Round 4: the CG systems are ASSEMBLED by the reference's own code too — eigen_utils::diagMat / diagSpMat / catSpMat of
src/utils/eigen_utils.cpp (compiled unmodified into libref_eigen.so) in the order of model.cpp:805-821, from the blocks formNormalEq +
applyL2Reg export (ref_solve_normal_eq_cg); full_system_triplets below (Python) stays as the cross-check of that assembly.

Run in the authoring container only:   make -C oracle && python tests/golden/make_eigen_golden.py
The fixture holds inputs and Eigen's outputs; it is data, not source."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(HERE, ".."))
from oracle import oracle as O  # noqa: E402


def schur_system(ne, lam, skip):
    """S and rhs of model.cpp:784-789 from the oracle's dense blocks (numpy: only the INPUT of ldlt is being produced here)."""
    A11 = ne["A11"][skip:, skip:]; A12 = ne["A12"][skip:, :]; b1 = ne["b1"][skip:]
    A11m = A11 + lam * np.diag(np.diag(A11))
    P = ne["P"]
    Binv = np.zeros((2 * P, 2 * P))
    for i in range(P):
        Am = ne["A22"][i] + lam * np.diag(np.diag(ne["A22"][i]))
        Binv[2 * i:2 * i + 2, 2 * i:2 * i + 2] = np.linalg.inv(Am)
    W = A12 @ Binv
    return A11m - W @ A12.T, b1 - W @ ne["b2"]


def full_system_triplets(ne, lam, skip):
    """[A11m A12; A12^T A22m] of model.cpp:806-826 as triplets (zeros dropped like sparseView()) and b = [b1; b2]."""
    A11 = ne["A11"][skip:, skip:]; A12 = ne["A12"][skip:, :]; b1 = ne["b1"][skip:]
    n, P = A11.shape[0], ne["P"]
    A11m = A11 + lam * np.diag(np.diag(A11))
    rows, cols, vals = [], [], []
    r, c = np.nonzero(A11m); rows += list(r); cols += list(c); vals += list(A11m[r, c])
    r, c = np.nonzero(A12); rows += list(r); cols += list(c + n); vals += list(A12[r, c])
    rows += list(c + n); cols += list(r); vals += list(A12[r, c])
    for i in range(P):
        Am = ne["A22"][i] + lam * np.diag(np.diag(ne["A22"][i]))
        for a in range(2):
            for b in range(2):
                rows.append(n + 2 * i + a); cols.append(n + 2 * i + b); vals.append(Am[a, b])
    return n + 2 * P, np.array(rows, np.int32), np.array(cols, np.int32), np.array(vals), np.concatenate([b1, ne["b2"]])


def unobserved_pose_workload():
    """K = 8 control poses but events only in the first 55 % of the window: the last control poses are constrained by no event."""
    from helpers import small_workload
    from emba_amd.legm import EventPacket
    w = small_workload(n_events=20000, pano_h=128, K=8, sensor=(32, 24), focal=30.0)
    ev = w.events
    n = (int(ev.size() * 0.55) // 100) * 100
    w.events = EventPacket(ev.x[:n], ev.y[:n], ev.polarity[:n], ev.t_ns[:n])
    return w


def main():
    from helpers import oracle_run, small_workload
    assert O.ref_eigen() is not None, "oracle/_ref/libref_eigen.so not built (needs /root/reference)"
    out = {}
    rng = np.random.default_rng(20241022)

    # ---- LDLT ----
    w = small_workload(n_events=20000, pano_h=256, K=6, sensor=(64, 48), focal=60.0)
    ne = oracle_run(O, w, dense_A12=True)["ne"]
    cases = {"spd": schur_system(ne, 1e-3, 3), "spd_full": schur_system(ne, 1e-2, 0)}
    wu = unobserved_pose_workload()
    neu = oracle_run(O, wu, dense_A12=True)["ne"]
    cases["unobserved_pose"] = schur_system(neu, 1e-3, 3)
    assert (np.diag(cases["unobserved_pose"][0]) == 0).sum() >= 3, "the case is meant to contain a control pose no event constrains"
    S, r = schur_system(ne, 1e-3, 3)
    S = S.copy(); S[4, 4] = -S[4, 4]
    cases["indefinite"] = (S, r)
    cases["zero"] = (np.zeros((6, 6)), np.arange(6.0))
    cases["one"] = (np.array([[2.5]]), np.array([1.0]))
    names = []
    for name, (S, rhs) in cases.items():
        x, d, tr, info = O.ref_ldlt_solve(S, rhs)
        out[f"ldlt_{name}_S"] = S; out[f"ldlt_{name}_rhs"] = rhs; out[f"ldlt_{name}_x"] = x; out[f"ldlt_{name}_D"] = d
        out[f"ldlt_{name}_tr"] = tr; out[f"ldlt_{name}_info"] = np.int32(info)
        names.append(name)
    out["ldlt_names"] = np.array(names)

    # ---- Matrix2d::inverse ----
    A = []
    for _ in range(64):
        v = rng.normal(size=(2, 2)); m = v @ v.T + 1e-3 * np.eye(2)
        A.append(m.ravel())
    A += [np.array([1.0, 2.0, 2.0, 4.0]), np.array([1e-3, 0.0, 0.0, 0.0]), np.array([0.0, 0.0, 0.0, 0.0]), np.array([3.0, 0.0, 0.0, 5.0])]
    A = np.array(A)
    out["inv2_A"] = A
    out["inv2_out"] = np.array([O.ref_inverse2(a).ravel() for a in A])

    # ---- ConjugateGradient ----
    wc = small_workload(n_events=2400, pano_h=64, K=5, sensor=(12, 8), focal=10.0)
    nec = oracle_run(O, wc, dense_A12=True)["ne"]
    for name, lam, skip in (("trimmed", 1e-3, 3), ("full", 1e-2, 0)):
        # the blocks as solver.cpp hands them to solveNormalEqCG (first-window trim :156-165 when skip = 3)
        A11 = nec["A11"][skip:, skip:]; A12 = nec["A12"][skip:, :]; b1 = nec["b1"][skip:]
        x1, x2, it, err, (rows, cols, vals) = O.ref_solve_normal_eq_cg(A11, A12, nec["A22"], b1, nec["b2"], lam)
        x = np.concatenate([x1, x2]); b = np.concatenate([b1, nec["b2"]]); n = x.size
        # cross-check 1: the Python assembly gives the same matrix (as a dense array: triplet order is Eigen's column-major one)
        n_py, r_py, c_py, v_py, b_py = full_system_triplets(nec, lam, skip)
        M_ref = np.zeros((n, n)); np.add.at(M_ref, (rows, cols), vals)
        M_py = np.zeros((n, n)); np.add.at(M_py, (r_py, c_py), v_py)
        assert n_py == n and np.array_equal(M_ref, M_py) and np.array_equal(b, b_py), "the reference's catSpMat assembly differs from the Python one"
        # cross-check 2: Eigen's CG on the triplets alone reproduces the end-to-end call bit for bit
        xt, itt, errt = O.ref_cg_solve(n, rows, cols, vals, b, 100, 1e-6)
        assert itt == it and np.array_equal(xt, x) and errt == err
        out[f"cg_{name}_rows"] = rows; out[f"cg_{name}_cols"] = cols; out[f"cg_{name}_vals"] = vals; out[f"cg_{name}_b"] = b
        out[f"cg_{name}_x"] = x; out[f"cg_{name}_iters"] = np.int32(it); out[f"cg_{name}_err"] = err
        out[f"cg_{name}_lam"] = lam; out[f"cg_{name}_skip"] = np.int32(skip)
        out[f"cg_{name}_A11"] = A11; out[f"cg_{name}_A12"] = A12; out[f"cg_{name}_A22"] = nec["A22"]; out[f"cg_{name}_b1"] = b1; out[f"cg_{name}_b2"] = nec["b2"]
    path = os.path.join(HERE, "eigen_solvers.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", {k: (int(out[f'cg_{k}_iters']), float(out[f'cg_{k}_err'])) for k in ("trimmed", "full")})


if __name__ == "__main__":
    main()
