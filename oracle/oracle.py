"""oracle/oracle.py — TEST INFRASTRUCTURE. NOT PART OF THE PRODUCT.

ctypes bindings for the CPU checker libraries:
  * oracle/libemba_oracle.so      — plain-C restatement of the reference hot path (emba_oracle.h)
  * oracle/_ref/libref_basalt.so  — the reference's own basalt/Sophus/Eigen headers (ref_basalt.cpp),
                                    present only where it was built (this container) or shipped prebuilt.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (emba_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.environ.get("EMBA_ORACLE_LIB", os.path.join(_HERE, "libemba_oracle.so"))   # EMBA_ORACLE_LIB: another build of the same source (bench.py: -march=native on the GPU box's host)
_REF = os.path.join(_HERE, "_ref", "libref_basalt.so")

_dp = C.POINTER(C.c_double)
_i32p = C.POINTER(C.c_int32)


def build():
    """Compile the checker libraries (gcc/g++ only; no GPU toolchain involved)."""
    import sys
    subprocess.check_call(["make", "-s", "-C", _HERE], stdout=sys.stderr)      # (stdout belongs to the caller: bench.py prints ONE JSON line there)


def _ptr(a, ty):
    return None if a is None else a.ctypes.data_as(ty)


class _Dump(C.Structure):
    _fields_ = [("pm", _dp), ("D", _dp), ("cp_idx", _i32p), ("inlier_idx", _i32p), ("pm_int", _i32p),
                ("dp", _dp), ("Gpm", _dp), ("temp", _dp), ("prev", _i32p)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        srcs = [os.path.join(_HERE, f) for f in ("emba_oracle.c", "emba_oracle.h", "Makefile")]
        if "EMBA_ORACLE_LIB" not in os.environ and (not os.path.exists(_LIB) or any(os.path.getmtime(f) > os.path.getmtime(_LIB) for f in srcs)):
            build()
        L = C.CDLL(_LIB)
        L.emba_oracle_create.restype = C.c_void_p
        L.emba_oracle_create.argtypes = [C.c_int] * 4 + [_dp, C.c_double]
        L.emba_oracle_destroy.argtypes = [C.c_void_p]
        L.emba_oracle_set_first_counted.argtypes = [C.c_void_p, C.c_size_t]
        L.emba_oracle_batch_mid_ns.restype = C.c_int64
        L.emba_oracle_batch_mid_ns.argtypes = [C.c_int64, C.c_int64]
        L.emba_oracle_spline_eval.restype = C.c_int
        L.emba_oracle_spline_eval.argtypes = [_dp, C.c_int, C.c_int64, C.c_int64, C.c_int64, _dp, _dp,
                                              C.POINTER(C.c_int), _dp]
        L.emba_oracle_so3_exp.argtypes = [_dp, _dp]
        L.emba_oracle_so3_log.argtypes = [_dp, _dp]
        L.emba_oracle_left_jacobian.argtypes = [_dp, _dp, _dp]
        L.emba_oracle_project.argtypes = [C.c_int, C.c_int, _dp, _dp, _dp]
        L.emba_oracle_warp.argtypes = [C.c_void_p, C.c_int, C.c_int, _dp, _dp, _dp]
        L.emba_oracle_hessian.argtypes = [_dp, _dp, C.c_int, C.c_int, _dp, _dp, _dp]
        L.emba_oracle_eval_data_error.restype = C.c_long
        L.emba_oracle_eval_data_error.argtypes = [C.c_void_p, _dp, C.c_int, C.c_int64, C.c_int64, _dp, _dp,
                                                  C.POINTER(C.c_uint16), C.POINTER(C.c_uint16),
                                                  C.POINTER(C.c_uint8), C.POINTER(C.c_int64), C.c_size_t,
                                                  _dp, _i32p, C.POINTER(_Dump)]
        L.emba_oracle_form_normal_eq.restype = C.c_long
        L.emba_oracle_form_normal_eq.argtypes = [C.c_void_p, _dp, C.c_int, _i32p, C.c_int, C.c_int,
                                                 C.c_double, _dp, _dp, C.POINTER(C.c_uint32), _dp, _dp, _dp]
        L.emba_oracle_apply_l2.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32), C.c_double, _dp,
                                           _dp, _dp, _dp]
        L.emba_oracle_update_map.argtypes = [C.c_size_t, C.POINTER(C.c_uint32), C.c_size_t, _dp, C.c_double, _dp, _dp]
        L.emba_oracle_solve_normal_eq.restype = C.c_int
        L.emba_oracle_solve_normal_eq.argtypes = [C.c_int, C.c_size_t, _dp, _dp, _dp, _dp, _dp, C.c_double, _dp, _dp]
        L.emba_oracle_data_cost.restype = C.c_double
        L.emba_oracle_data_cost.argtypes = [_dp, C.c_size_t, C.c_int, C.c_double]
        L.emba_oracle_reg_cost.restype = C.c_double
        L.emba_oracle_reg_cost.argtypes = [_dp, _dp, C.c_size_t, C.c_double]
        _sp = [C.c_void_p, _dp, C.c_int, _i32p, C.c_int, C.c_int, C.c_double, _dp, _dp, C.c_size_t, C.POINTER(C.c_uint32), _dp, _dp,
               C.c_double, C.c_int]
        L.emba_oracle_solve_sparse.restype = C.c_int
        L.emba_oracle_solve_sparse.argtypes = _sp + [_dp, _dp]
        L.emba_oracle_solve_cg_sparse.restype = C.c_int
        L.emba_oracle_solve_cg_sparse.argtypes = _sp + [C.c_int, C.c_double, _dp, _dp, C.POINTER(C.c_int), _dp]
        L.emba_oracle_count_map.restype = C.c_long
        L.emba_oracle_count_map.argtypes = [C.c_void_p, _dp, C.c_int, C.c_int64, C.c_int64, C.POINTER(C.c_uint16),
                                            C.POINTER(C.c_uint16), C.POINTER(C.c_int64), C.c_size_t, _dp, _i32p, _i32p]
        L.emba_oracle_inverse2.argtypes = [_dp, _dp]
        L.emba_oracle_ldlt_solve.restype = C.c_int
        L.emba_oracle_ldlt_solve.argtypes = [_dp, C.c_int, _dp, _dp, _dp, _i32p]
        L.emba_oracle_set_threads.argtypes = [C.c_int]
        L.emba_oracle_get_threads.restype = C.c_int
        L.emba_oracle_max_threads.restype = C.c_int
        _lib = L
    return _lib


_ref = None


def ref_available():
    return os.path.exists(_REF)


def ref():
    """The reference's own basalt So3Spline<2> (oracle/_ref), or None when it was not built/shipped."""
    global _ref
    if _ref is None and ref_available():
        R = C.CDLL(_REF)
        R.ref_so3spline2_evaluate.restype = C.c_int
        R.ref_so3spline2_evaluate.argtypes = [_dp, C.c_int, C.c_int64, C.c_int64, C.c_int64, _dp, _dp,
                                              C.POINTER(C.c_int), _dp]
        R.ref_so3_exp.argtypes = [_dp, _dp]
        R.ref_so3_log.argtypes = [_dp, _dp]
        R.ref_left_jacobian.argtypes = [_dp, _dp, _dp]
        _ref = R
    return _ref


_ref_eigen = None
_REF_EIGEN = os.path.join(_HERE, "_ref", "libref_eigen.so")


def ref_eigen():
    """The reference's own vendored Eigen behind oracle/ref_eigen.cpp (LDLT, Matrix2d::inverse, ConjugateGradient), or None when
    oracle/_ref was not built / shipped."""
    global _ref_eigen
    if _ref_eigen is None and os.path.exists(_REF_EIGEN):
        R = C.CDLL(_REF_EIGEN)
        R.ref_ldlt_solve.restype = C.c_int
        R.ref_ldlt_solve.argtypes = [C.c_int, _dp, _dp, _dp, _dp, _i32p]
        R.ref_inverse2.argtypes = [_dp, _dp]
        R.ref_cg_solve.argtypes = [C.c_int, C.c_long, _i32p, _i32p, _dp, _dp, C.c_int, C.c_double, _dp, C.POINTER(C.c_int), _dp]
        _ref_eigen = R
    return _ref_eigen


def ref_ldlt_solve(S, rhs):
    """Eigen: S.ldlt().solve(rhs) -> (x, vecD, transpositions, info)."""
    S = np.array(S, dtype=np.float64, order="F"); rhs = _f64(rhs)
    n = S.shape[0]
    x = np.zeros(n); d = np.zeros(n); tr = np.zeros(n, dtype=np.int32)
    info = ref_eigen().ref_ldlt_solve(n, _ptr(S, _dp), _ptr(rhs, _dp), _ptr(x, _dp), _ptr(d, _dp), _ptr(tr, _i32p))
    return x, d, tr, info


def ref_inverse2(A):
    A = _f64(np.asarray(A, dtype=np.float64).reshape(4)); out = np.zeros(4)
    ref_eigen().ref_inverse2(_ptr(A, _dp), _ptr(out, _dp))
    return out.reshape(2, 2)


def ref_cg_solve(n, rows, cols, vals, b, max_iter=100, tol=1e-6):
    """Eigen::ConjugateGradient<SpMat, Lower|Upper> on the matrix given by triplets -> (x, iterations, error)."""
    rows = np.ascontiguousarray(rows, dtype=np.int32); cols = np.ascontiguousarray(cols, dtype=np.int32); vals = _f64(vals); b = _f64(b)
    x = np.zeros(n); it = C.c_int(0); err = np.zeros(1)
    ref_eigen().ref_cg_solve(int(n), int(vals.size), _ptr(rows, _i32p), _ptr(cols, _i32p), _ptr(vals, _dp), _ptr(b, _dp), int(max_iter), float(tol),
                             _ptr(x, _dp), C.byref(it), _ptr(err, _dp))
    return x, it.value, float(err[0])


def ref_solve_normal_eq_cg(A11, A12, A22, b1, b2, lam):
    """LEGM::solveNormalEqCG (model.cpp:794-840) end to end by the reference's own code: eigen_utils::diagMat / diagSpMat / catSpMat
    (src/utils/eigen_utils.cpp, compiled unmodified) + Eigen's ConjugateGradient.  A11 n x n, A12 n x 2P (dense), A22 P x 2 x 2.
    Returns (x1, x2, iterations, error, (rows, cols, vals) of the assembled system)."""
    A11 = np.asfortranarray(A11, dtype=np.float64); A12 = np.asfortranarray(A12, dtype=np.float64)
    n = A11.shape[0]; P = A12.shape[1] // 2
    A22 = _f64(A22).reshape(P, 4); b1 = _f64(b1); b2 = _f64(b2)
    x1 = np.zeros(n); x2 = np.zeros(max(2 * P, 1)); it = C.c_int(0); err = np.zeros(1)
    cap = n * n + 2 * n * 2 * P + 4 * P + 8
    rows = np.zeros(cap, np.int32); cols = np.zeros(cap, np.int32); vals = np.zeros(cap); nnz = C.c_long(0)
    f = ref_eigen().ref_solve_normal_eq_cg
    f.restype = None
    f(C.c_int(n), C.c_long(P), A11.ctypes.data_as(_dp), A12.ctypes.data_as(_dp), _ptr(A22, _dp), _ptr(b1, _dp), _ptr(b2, _dp), C.c_double(lam),
      _ptr(x1, _dp), _ptr(x2, _dp), C.byref(it), _ptr(err, _dp), C.c_long(cap), _ptr(rows, _i32p), _ptr(cols, _i32p), _ptr(vals, _dp), C.byref(nnz))
    k = nnz.value
    assert k <= cap
    return x1, x2[: 2 * P], it.value, float(err[0]), (rows[:k].copy(), cols[:k].copy(), vals[:k].copy())


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def set_threads(n):
    """1 = "ref" mode (one thread, the reference's order: the checker); n > 1 = "omp" CPU-baseline mode."""
    lib().emba_oracle_set_threads(int(n))


def max_threads():
    return int(lib().emba_oracle_max_threads())


def spline_eval(knots_xyzw, t0_ns, dt_ns, t_ns, use_ref=False):
    """Returns (q_xyzw[4], R[3,3], cp_idx, J36[3,6]) or None if t is outside the knots."""
    knots = _f64(knots_xyzw).reshape(-1, 4)
    q = np.zeros(4); R = np.zeros(9); J = np.zeros(18); s = C.c_int(0)
    fn = ref().ref_so3spline2_evaluate if use_ref else lib().emba_oracle_spline_eval
    rc = fn(_ptr(knots, _dp), knots.shape[0], int(t0_ns), int(dt_ns), int(t_ns), _ptr(q, _dp), _ptr(R, _dp),
            C.byref(s), _ptr(J, _dp))
    if rc != 0:
        return None
    return q, R.reshape(3, 3), s.value, J.reshape(3, 6)


def so3_exp(w, use_ref=False):
    w = _f64(w); q = np.zeros(4)
    (ref().ref_so3_exp if use_ref else lib().emba_oracle_so3_exp)(_ptr(w, _dp), _ptr(q, _dp))
    return q


def so3_log(q, use_ref=False):
    q = _f64(q); w = np.zeros(3)
    (ref().ref_so3_log if use_ref else lib().emba_oracle_so3_log)(_ptr(q, _dp), _ptr(w, _dp))
    return w


def left_jacobian(phi, use_ref=False):
    phi = _f64(phi); J = np.zeros(9); Ji = np.zeros(9)
    (ref().ref_left_jacobian if use_ref else lib().emba_oracle_left_jacobian)(_ptr(phi, _dp), _ptr(J, _dp),
                                                                             _ptr(Ji, _dp))
    return J.reshape(3, 3), Ji.reshape(3, 3)


def batch_mid_ns(t_first, t_last):
    return lib().emba_oracle_batch_mid_ns(int(t_first), int(t_last))


def project(pano_w, pano_h, rb):
    rb = _f64(rb); pm = np.zeros(2); J = np.zeros(6)
    lib().emba_oracle_project(pano_w, pano_h, _ptr(rb, _dp), _ptr(pm, _dp), _ptr(J, _dp))
    return pm, J.reshape(2, 3)


def hessian(Gx, Gy):
    Gx = _f64(Gx); Gy = _f64(Gy)
    H, W = Gx.shape
    out = [np.zeros((H, W)) for _ in range(3)]
    lib().emba_oracle_hessian(_ptr(Gx, _dp), _ptr(Gy, _dp), H, W, *[_ptr(o, _dp) for o in out])
    return out


def update_map(active, x2, damping, Gx, Gy):
    """LEGM::updateMap (model.cpp:863-903) on copies; returns (Gx_new, Gy_new)."""
    Gx = _f64(Gx).copy(); Gy = _f64(Gy).copy()
    act = np.ascontiguousarray(active, dtype=np.uint32); x2 = _f64(x2)
    lib().emba_oracle_update_map(act.size, _ptr(act, C.POINTER(C.c_uint32)), Gx.size, _ptr(x2, _dp), float(damping), _ptr(Gx, _dp), _ptr(Gy, _dp))
    return Gx, Gy


def solve_normal_eq(ne, lam, fix_first_pose=False):
    """LEGM::solveNormalEq (model.cpp:721-792) on a dict from OracleLEGM.form_normal_eq(dense_A12=True) (after apply_l2).
    fix_first_pose trims rows/cols 0..2 like solver.cpp:156-165; x1 comes back with zeros there."""
    A11 = np.asfortranarray(ne["A11"]); A12 = np.asfortranarray(ne["A12"]); b1 = _f64(ne["b1"])
    sk = 3 if fix_first_pose else 0
    A11s = np.asfortranarray(A11[sk:, sk:]); A12s = np.asfortranarray(A12[sk:, :]); b1s = _f64(b1[sk:])
    n, P = A11s.shape[0], ne["P"]
    A22 = _f64(ne["A22"]); b2 = _f64(ne["b2"])
    x1 = np.zeros(n); x2 = np.zeros(2 * max(P, 1))
    rc = lib().emba_oracle_solve_normal_eq(n, P, _ptr(A11s, _dp), _ptr(A12s, _dp), _ptr(A22, _dp), _ptr(b1s, _dp), _ptr(b2, _dp), float(lam),
                                           _ptr(x1, _dp), _ptr(x2, _dp))
    # (rc = what Eigen's ldlt.info() would report; the reference never reads it and uses x1 / x2 as they come)
    return np.concatenate([np.zeros(sk), x1]), x2[:2 * P]


def ldlt_solve(S, rhs):
    """x = S.ldlt().solve(rhs) (model.cpp:789; Eigen's pivoted LDLT restated): returns (x, vecD, transpositions, info)."""
    S = np.array(S, dtype=np.float64, order="F"); rhs = _f64(rhs)
    n = S.shape[0]
    x = np.zeros(n); d = np.zeros(n); tr = np.zeros(n, dtype=np.int32)
    info = lib().emba_oracle_ldlt_solve(_ptr(S, _dp), n, _ptr(rhs, _dp), _ptr(x, _dp), _ptr(d, _dp), _ptr(tr, _i32p))
    return x, d, tr, info


def inverse2(A):
    """Eigen::Matrix2d::inverse() (model.cpp:750) restated."""
    A = _f64(np.asarray(A, dtype=np.float64).reshape(4)); out = np.zeros(4)
    lib().emba_oracle_inverse2(_ptr(A, _dp), _ptr(out, _dp))
    return out.reshape(2, 2)


def data_cost(ep, irls=0, a=0.0):
    ep = _f64(ep)
    return lib().emba_oracle_data_cost(_ptr(ep, _dp), ep.size, irls, a)


def reg_cost(Gx, Gy, alpha):
    Gx = _f64(Gx); Gy = _f64(Gy)
    return lib().emba_oracle_reg_cost(_ptr(Gx, _dp), _ptr(Gy, _dp), Gx.size, alpha)


class OracleLEGM:
    """Mirror of EMBA::LEGM (include/emba/model.h:72-133) on plain numpy arrays, CPU oracle inside."""

    def __init__(self, sensor_w, sensor_h, pano_w, pano_h, bearing_lut, C_th):
        self.sw, self.sh, self.W, self.H = sensor_w, sensor_h, pano_w, pano_h
        self.lut = _f64(bearing_lut).reshape(sensor_w * sensor_h, 3)
        self._o = lib().emba_oracle_create(sensor_w, sensor_h, pano_w, pano_h, _ptr(self.lut, _dp), C_th)
        self.n = 0

    def __del__(self):
        if getattr(self, "_o", None):
            lib().emba_oracle_destroy(self._o)
            self._o = None

    def warp(self, x, y, q_xyzw):
        q = _f64(q_xyzw); pm = np.zeros(2); J = np.zeros(6)
        lib().emba_oracle_warp(self._o, int(x), int(y), _ptr(q, _dp), _ptr(pm, _dp), _ptr(J, _dp))
        return pm, J.reshape(2, 3)

    def evaluate_data_error(self, knots_xyzw, t0_ns, dt_ns, Gx, Gy, x, y, pol, t_ns, dump=False, first_counted=0):
        """first_counted: shard view — events before this index only serve as predecessors (emba_oracle_set_first_counted)."""
        lib().emba_oracle_set_first_counted(self._o, int(first_counted))
        knots = _f64(knots_xyzw).reshape(-1, 4)
        Gx = _f64(Gx); Gy = _f64(Gy)
        x = np.ascontiguousarray(x, dtype=np.uint16); y = np.ascontiguousarray(y, dtype=np.uint16)
        pol = np.ascontiguousarray(pol, dtype=np.uint8); t_ns = np.ascontiguousarray(t_ns, dtype=np.int64)
        n = x.size
        self.n = n
        ep = np.zeros(max(n, 1))
        num_ev_map = np.zeros((self.H, self.W), dtype=np.int32)
        d = None
        dstruct = None
        if dump:
            d = dict(pm=np.zeros((n, 2)), D=np.zeros((n, 2, 6)), cp_idx=np.zeros(n, np.int32),
                     inlier_idx=np.zeros(n, np.int32), pm_int=np.full((n, 2), -1, np.int32),
                     dp=np.zeros((n, 2)), Gpm=np.zeros((n, 2)), temp=np.zeros((n, 2)),
                     prev=np.zeros(n, np.int32))
            dstruct = _Dump(_ptr(d["pm"], _dp), _ptr(d["D"], _dp), _ptr(d["cp_idx"], _i32p),
                            _ptr(d["inlier_idx"], _i32p), _ptr(d["pm_int"], _i32p), _ptr(d["dp"], _dp),
                            _ptr(d["Gpm"], _dp), _ptr(d["temp"], _dp), _ptr(d["prev"], _i32p))
        m = lib().emba_oracle_eval_data_error(
            self._o, _ptr(knots, _dp), knots.shape[0], int(t0_ns), int(dt_ns), _ptr(Gx, _dp), _ptr(Gy, _dp),
            _ptr(x, C.POINTER(C.c_uint16)), _ptr(y, C.POINTER(C.c_uint16)), _ptr(pol, C.POINTER(C.c_uint8)),
            _ptr(t_ns, C.POINTER(C.c_int64)), n, _ptr(ep, _dp), _ptr(num_ev_map, _i32p),
            C.byref(dstruct) if dstruct is not None else None)
        if m < 0:
            raise ValueError("batch time outside the spline's knots")
        ep = ep[:m].copy()
        return (ep, num_ev_map, d) if dump else (ep, num_ev_map)

    def count_map(self, knots_xyzw, t0_ns, dt_ns, x, y, t_ns, want_pm=False, want_pm_int=False):
        """Index-level results only (emba_oracle_count_map): (n_inliers, num_ev_map[, pm][, pm_int])."""
        knots = _f64(knots_xyzw).reshape(-1, 4)
        x = np.ascontiguousarray(x, dtype=np.uint16); y = np.ascontiguousarray(y, dtype=np.uint16)
        t_ns = np.ascontiguousarray(t_ns, dtype=np.int64)
        n = x.size
        nem = np.zeros((self.H, self.W), dtype=np.int32)
        pm = np.zeros((n, 2)) if want_pm else None
        pmi = np.zeros((n, 2), dtype=np.int32) if want_pm_int else None
        m = lib().emba_oracle_count_map(self._o, _ptr(knots, _dp), knots.shape[0], int(t0_ns), int(dt_ns),
                                        _ptr(x, C.POINTER(C.c_uint16)), _ptr(y, C.POINTER(C.c_uint16)),
                                        _ptr(t_ns, C.POINTER(C.c_int64)), n, _ptr(pm, _dp), _ptr(nem, _i32p), _ptr(pmi, _i32p))
        if m < 0:
            raise ValueError("batch time outside the spline's knots")
        out = [int(m), nem]
        if want_pm:
            out.append(pm)
        if want_pm_int:
            out.append(pmi)
        return tuple(out)

    def form_normal_eq(self, ep, K, num_ev_map, thres, irls=0, a=0.0, dense_A12=False):
        ep = _f64(ep)
        nem = np.ascontiguousarray(num_ev_map, dtype=np.int32)
        npix = self.W * self.H
        A11 = np.zeros((3 * K, 3 * K), order="F"); b1 = np.zeros(3 * K)
        active = np.zeros(npix, dtype=np.uint32)
        P_guess = int((nem >= thres).sum())
        A22 = np.zeros((max(P_guess, 1), 2, 2)); b2 = np.zeros(2 * max(P_guess, 1))
        A12 = np.zeros((3 * K, 2 * P_guess), order="F") if dense_A12 else None
        P = lib().emba_oracle_form_normal_eq(self._o, _ptr(ep, _dp), K, _ptr(nem, _i32p), thres, irls, a,
                                             _ptr(A11, _dp), _ptr(b1, _dp), _ptr(active, C.POINTER(C.c_uint32)),
                                             _ptr(A22, _dp), _ptr(b2, _dp), _ptr(A12, _dp))
        assert P == P_guess
        return dict(A11=A11, b1=b1, active=active[:P].copy(), A22=A22[:P], b2=b2[:2 * P], A12=A12, P=P)

    def _sparse_args(self, ne, ep, K, num_ev_map, thres, irls, a, lam, fix_first_pose):
        ep = _f64(ep); nem = np.ascontiguousarray(num_ev_map, dtype=np.int32)
        A11 = np.asfortranarray(ne["A11"]); b1 = _f64(ne["b1"]); A22 = _f64(ne["A22"]); b2 = _f64(ne["b2"])
        act = np.ascontiguousarray(ne["active"], dtype=np.uint32)
        keep = (ep, nem, A11, b1, A22, b2, act)
        return keep, [self._o, _ptr(ep, _dp), K, _ptr(nem, _i32p), int(thres), int(irls), float(a), _ptr(A11, _dp), _ptr(b1, _dp),
                      ne["P"], _ptr(act, C.POINTER(C.c_uint32)), _ptr(A22, _dp), _ptr(b2, _dp), float(lam), 3 if fix_first_pose else 0]

    def solve_sparse(self, ne, ep, K, num_ev_map, thres, irls=0, a=0.0, lam=1e-3, fix_first_pose=False):
        """LEGM::solveNormalEq (model.cpp:721-792) from the sparse A12 factors of the last evaluate_data_error (after apply_l2)."""
        keep, args = self._sparse_args(ne, ep, K, num_ev_map, thres, irls, a, lam, fix_first_pose)
        x1 = np.zeros(3 * K); x2 = np.zeros(2 * max(ne["P"], 1))
        lib().emba_oracle_solve_sparse(*args, _ptr(x1, _dp), _ptr(x2, _dp))     # (returns ldlt.info(); the reference never reads it)
        return x1, x2[:2 * ne["P"]]

    def solve_cg_sparse(self, ne, ep, K, num_ev_map, thres, irls=0, a=0.0, lam=1e-3, fix_first_pose=False, max_iter=100, tol=1e-6):
        """LEGM::solveNormalEqCG (model.cpp:794-840): returns (x1, x2, iterations, error)."""
        keep, args = self._sparse_args(ne, ep, K, num_ev_map, thres, irls, a, lam, fix_first_pose)
        x1 = np.zeros(3 * K); x2 = np.zeros(2 * max(ne["P"], 1)); it = C.c_int(0); err = np.zeros(1)
        lib().emba_oracle_solve_cg_sparse(*args, int(max_iter), float(tol), _ptr(x1, _dp), _ptr(x2, _dp), C.byref(it), _ptr(err, _dp))
        return x1, x2[:2 * ne["P"]], it.value, float(err[0])

    def apply_l2(self, ne, alpha, Gx, Gy):
        Gx = _f64(Gx); Gy = _f64(Gy)
        if ne["P"] == 0:
            return ne
        lib().emba_oracle_apply_l2(self._o, ne["P"], _ptr(ne["active"], C.POINTER(C.c_uint32)), alpha,
                                   _ptr(Gx, _dp), _ptr(Gy, _dp), _ptr(ne["A22"], _dp), _ptr(ne["b2"], _dp))
        return ne
