"""Host-side mirror of the reference's measurement-model interface `EMBA::LEGM`
(reference include/emba/model.h:72-133) on plain numpy arrays, calling the HIP library through the
C ABI of include/emba_hip.h.  Method names, argument meaning and error behaviour follow the reference:

    LEGM(camera_info, C_th, pano_width, pano_height)        model.h:76-77   -> LEGM(sensor_w, sensor_h, bearing_lut, ...)
    evaluateDataError(traj, Gx, Gy, events, eval_deriv, num_ev_map)  :83-84
    evaluateRegError / evaluateRobustDataCost                :87, :90       -> regCost / dataCost (scalar, on device)
    formNormalEq / formNormalEqIRLS                          :93-103
    applyL2Reg                                               :106-108

Differences forced by the boundary: the bearing-vector LUT is an input (the reference
builds it from ROS camera_info, event_pano_warper.cpp:27-41); the trajectory is passed as its control
quaternions + (t0_ns, dt_ns); events are registered once per window with set_events() because the per-pixel
event lists are pose-independent.  Invariant violations raise EmbaError (the reference aborts via glog CHECK).
"""
import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib
from ._lib import EmbaCfg, EmbaError

_dp, _i32p, _u32p = _lib._dp, _lib._i32p, _lib._u32p
_u16p, _u8p, _i64p = _lib._u16p, _lib._u8p, _lib._i64p

COST_TYPES = {"quadratic": 0, "huber": 1, "cauchy": 2}


def _p(a, ty):
    return None if a is None else a.ctypes.data_as(ty)


@dataclass
class LinearTrajectory:
    """The data `LinearTrajectory` hands to the hot path (reference include/utils/trajectory.h:109-190):
    control rotations as unit quaternions (x,y,z,w) and the spline timing of trajectory.cpp:59-64
    (t_beg_ns = int64(1e9*t_beg), dt_ns = int64(1e9*dt_knots))."""
    knots_xyzw: np.ndarray
    t0_ns: int
    dt_ns: int

    @classmethod
    def from_seconds(cls, t_beg, dt_knots, knots_xyzw):
        return cls(np.ascontiguousarray(knots_xyzw, dtype=np.float64).reshape(-1, 4), int(1e9 * t_beg), int(1e9 * dt_knots))

    def size(self):
        return self.knots_xyzw.shape[0]


@dataclass
class EventPacket:
    """std::vector<dvs_msgs::Event> (model.h:17) as a struct of arrays; ts as int64 nanoseconds."""
    x: np.ndarray
    y: np.ndarray
    polarity: np.ndarray
    t_ns: np.ndarray

    def size(self):
        return int(self.x.size)


class LEGM:
    def __init__(self, sensor_w, sensor_h, bearing_lut, C_th, pano_width, pano_height, device=0, stream=None):
        self._L = _lib.load()
        self._ctx = C.c_void_p()
        self.sensor_w, self.sensor_h = int(sensor_w), int(sensor_h)
        self.W, self.H = int(pano_width), int(pano_height)
        lut = np.ascontiguousarray(bearing_lut, dtype=np.float64).reshape(self.sensor_w * self.sensor_h, 3)
        cfg = EmbaCfg(self.sensor_w, self.sensor_h, self.W, self.H, _p(lut, _dp), float(C_th), 100, 10.0, int(device),
                      C.c_void_p(stream) if stream else None)
        st = self._L.emba_create(C.byref(cfg), C.byref(self._ctx))
        if st != _lib.OK:
            self._ctx = C.c_void_p()
            raise EmbaError(st, self._L.emba_last_error(None).decode())
        self.stream = int(stream) if stream else None   # the caller's HIP stream (None: the library made its own)
        self.n_events = 0
        self.K = 0
        self._P = 0
        self._keep = []  # arrays that must outlive an asynchronous launch

    # -- lifetime ------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_ctx", None) and self._ctx.value:
            self._L.emba_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, st):
        if st != _lib.OK:
            raise EmbaError(st, self._L.emba_last_error(self._ctx).decode())

    # -- events (once per window) --------------------------------------------------------------
    def set_events(self, events, halo=None):
        """events: EventPacket sorted by time.  halo: optional (x, y, batch_t_ns) arrays (multi-GPU shards)."""
        x = np.ascontiguousarray(events.x, dtype=np.uint16)
        y = np.ascontiguousarray(events.y, dtype=np.uint16)
        pol = np.ascontiguousarray(events.polarity, dtype=np.uint8)
        t = np.ascontiguousarray(events.t_ns, dtype=np.int64)
        if not (x.size == y.size == pol.size == t.size):
            raise ValueError("event arrays differ in length")
        hx = hy = ht = None
        nh = 0
        if halo is not None and len(halo[0]):
            hx = np.ascontiguousarray(halo[0], dtype=np.uint16)
            hy = np.ascontiguousarray(halo[1], dtype=np.uint16)
            ht = np.ascontiguousarray(halo[2], dtype=np.int64)
            nh = hx.size
        self._check(self._L.emba_set_events(self._ctx, _p(x, _u16p), _p(y, _u16p), _p(pol, _u8p), _p(t, _i64p), x.size,
                                            _p(hx, _u16p), _p(hy, _u16p), _p(ht, _i64p), nh))
        self.n_events = x.size

    def set_events_dev(self, x_ptr, y_ptr, pol_ptr, t_ptr, n, halo=None):
        """The same with the arrays already in HBM: device pointers to uint16 x, uint16 y, uint8 polarity, int64 t_ns (n each);
        halo = (hx_ptr, hy_ptr, hbt_ptr, n_halo) or None.  Nothing of the per-window structure is built on the host."""
        hx = hy = ht = None
        nh = 0
        if halo is not None and halo[3]:
            hx, hy, ht, nh = C.c_void_p(halo[0]), C.c_void_p(halo[1]), C.c_void_p(halo[2]), int(halo[3])
        self._check(self._L.emba_set_events_dev(self._ctx, C.c_void_p(x_ptr), C.c_void_p(y_ptr), C.c_void_p(pol_ptr), C.c_void_p(t_ptr), int(n),
                                                hx, hy, ht, nh))
        self.n_events = int(n)

    def setup_info(self):
        """Diagnostics of the once-per-window work (emba_last_setup_ms)."""
        a, b = C.c_double(0), C.c_double(0)
        t = C.c_int32(0)
        ne, nc = C.c_size_t(0), C.c_size_t(0)
        self._check(self._L.emba_last_setup_ms(self._ctx, C.byref(a), C.byref(b), C.byref(t), C.byref(ne), C.byref(nc)))
        pp, lf = C.c_double(0), C.c_double(0)
        self._check(self._L.emba_last_order_stats(self._ctx, C.byref(pp), C.byref(lf)))
        fi = C.c_double(0)
        self._check(self._L.emba_last_order_inlier_estimate(self._ctx, C.byref(fi)))
        g = [C.c_int32(0) for _ in range(5)]
        self._check(self._L.emba_last_tile_geometry(self._ctx, *[C.byref(v) for v in g]))
        return dict(set_events_ms=a.value, prepare_ms=b.value, tile_order=bool(t.value), entries=ne.value, chunks=nc.value,
                    events_per_pano_px=round(pp.value, 2), lead_in_frac=round(lf.value, 3), inlier_frac_predicted=round(fi.value, 3),
                    tile=dict(w=g[0].value, h=g[1].value, pitch_x=g[2].value, pitch_y=g[3].value, reserve=g[4].value) if t.value else None)

    def tile_drift(self):
        """(inliers of the last resolved evaluation outside their tile, times the window was re-binned): emba_last_tile_drift."""
        n, r = C.c_size_t(0), C.c_int32(0)
        self._check(self._L.emba_last_tile_drift(self._ctx, C.byref(n), C.byref(r)))
        return n.value, r.value

    def event_counts(self):
        a, b = C.c_size_t(0), C.c_size_t(0)
        self._check(self._L.emba_event_counts(self._ctx, C.byref(a), C.byref(b)))
        return a.value, b.value

    # -- reference-shaped one-shot interface ---------------------------------------------------
    def evaluateDataError(self, traj, Gx, Gy, events=None, eval_deriv=True, num_ev_map=None):
        """model.cpp:72-258.  Returns ep (inlier residuals, reference order); fills num_ev_map (int32 H x W)."""
        if events is not None:
            self.set_events(events)
        if Gx is None and Gy is None:      # evaluate on the map already resident on the device (uploaded, updated or bound)
            pass
        else:
            Gx = np.ascontiguousarray(Gx, dtype=np.float64)
            Gy = np.ascontiguousarray(Gy, dtype=np.float64)
            if Gx.shape != (self.H, self.W) or Gy.shape != (self.H, self.W):
                raise ValueError("Gx/Gy must be pano_height x pano_width float64")
        knots = np.ascontiguousarray(traj.knots_xyzw, dtype=np.float64).reshape(-1, 4)
        self.K = knots.shape[0]
        ep = np.empty(max(self.n_events, 1), dtype=np.float64)
        n_inl = C.c_size_t(0)
        if num_ev_map is None:
            num_ev_map = np.empty((self.H, self.W), dtype=np.int32)
        assert num_ev_map.dtype == np.int32 and num_ev_map.flags.c_contiguous and num_ev_map.shape == (self.H, self.W)
        self._check(self._L.emba_eval_data_error(self._ctx, _p(knots, _dp), self.K, int(traj.t0_ns), int(traj.dt_ns), _p(Gx, _dp),
                                                 _p(Gy, _dp), 1 if eval_deriv else 0, _p(ep, _dp), C.byref(n_inl),
                                                 _p(num_ev_map, _i32p)))
        self.num_ev_map = num_ev_map
        return ep[:n_inl.value].copy()

    def dataCost(self, cost_type="quadratic", a=0.0):
        """0.5*ep.dot(ep) (solver.cpp:88) or evaluateRobustDataCost (model.cpp:279-314), reduced on the device."""
        v = C.c_double(0)
        self._check(self._L.emba_data_cost(self._ctx, COST_TYPES[cost_type], float(a), C.byref(v)))
        return v.value

    def costs(self, cost_type="quadratic", a=0.0, alpha=0.0):
        """(dataCost, regCost) with one host synchronisation: emba_costs."""
        d, r = C.c_double(0), C.c_double(0)
        self._check(self._L.emba_costs(self._ctx, COST_TYPES[cost_type], float(a), float(alpha), C.byref(d), C.byref(r)))
        return d.value, r.value

    def regCost(self, alpha):
        """alpha*0.5*|evaluateRegError|^2 (model.cpp:260-277, solver.cpp:90), reduced on the device."""
        v = C.c_double(0)
        self._check(self._L.emba_reg_cost(self._ctx, float(alpha), C.byref(v)))
        return v.value

    def _form(self, ep, thres, irls, a, alpha, dense_A12):
        ep_h = None if ep is None else np.ascontiguousarray(ep, dtype=np.float64)
        P = C.c_size_t(0)
        pl = C.c_size_t(0)
        self._check(self._L.emba_form_active(self._ctx, int(thres), C.byref(P), C.byref(pl)))
        self._P = P.value
        self._check(self._L.emba_form_accumulate(self._ctx, _p(ep_h, _dp), irls, float(a)))
        return self._finish(alpha, dense_A12)

    def _finish(self, alpha, dense_A12):
        P, dim = self._P, 3 * self.K
        A11 = np.zeros((dim, dim), order="F"); b1 = np.zeros(dim)
        active = np.zeros(max(P, 1), dtype=np.uint32)
        A22 = np.zeros((max(P, 1), 2, 2)); b2 = np.zeros(2 * max(P, 1))
        A12 = np.zeros((dim, 2 * P), order="F") if dense_A12 else None
        self._check(self._L.emba_form_finish(self._ctx, float(alpha), _p(A11, _dp), _p(b1, _dp), _p(active, _u32p), max(P, 1),
                                             _p(A22, _dp), _p(b2, _dp), _p(A12, _dp) if (dense_A12 and P) else None))
        return dict(A11=A11, b1=b1, active=active[:P].copy(), A22=A22[:P], b2=b2[:2 * P], A12=A12, P=P)

    def formNormalEq(self, ep, num_ctrl_poses, num_ev_map=None, thres_valid_pixel=5, dense_A12=False):
        """model.cpp:316-491.  num_ev_map is accepted for signature parity; the device-resident copy produced by the
        last evaluateDataError is what is used (they are the same object in the reference's call pattern, solver.cpp:114-126)."""
        if num_ctrl_poses != self.K:
            raise EmbaError(_lib.ERR_INVALID_ARG, "num_ctrl_poses differs from the trajectory used in evaluateDataError")
        return self._form(ep, thres_valid_pixel, 0, 0.0, 0.0, dense_A12)

    def formNormalEqIRLS(self, ep, num_ctrl_poses, num_ev_map=None, thres_valid_pixel=5, cost_type="huber", a=0.1,
                         dense_A12=False):
        """model.cpp:493-687."""
        if num_ctrl_poses != self.K:
            raise EmbaError(_lib.ERR_INVALID_ARG, "num_ctrl_poses differs from the trajectory used in evaluateDataError")
        return self._form(ep, thres_valid_pixel, COST_TYPES[cost_type], a, 0.0, dense_A12)

    def applyL2Reg(self, alpha, dense_A12=False):
        """model.cpp:689-719, applied to the device-resident pack; returns the updated blocks (call once per formNormalEq)."""
        return self._finish(alpha, dense_A12)

    keeps_equations_on_reject = True   # rejectMap / rejectTrial make the equations formed before the trial current again (second record set)
    async_phases = True            # eval_finish(sync=False) / form_active(thres, sync=False): the LM loop's only synchronisations are costs() and the solve
    supports_resident_x2 = True    # solveNormalEq[CG](..., resident_x2=True) returns x2 = None; updateMap(None, damping) applies the device copy

    def solveNormalEq(self, lam, fix_first_pose=False, resident_x2=False):
        """model.cpp:721-792 on the device-resident, L2-regularised normal equations (call after applyL2Reg, solver.cpp:130,190):
        returns (x1 [3K, zeros for a fixed first pose], x2 [2P]).  resident_x2: x2 stays on the device (returned as None) for the
        updateMap(None, ...) that follows — the reference hands it from the solver to updateMap and nowhere else (solver.cpp:193-239)."""
        self.last_counts()     # (P may have been left on the device by form_active(sync=False): the output buffer is sized from it)
        x1 = np.zeros(3 * self.K)
        x2 = None if resident_x2 else np.zeros(2 * max(self._P, 1))
        self._check(self._L.emba_solve_normal_eq(self._ctx, float(lam), 1 if fix_first_pose else 0, _p(x1, _dp), None if resident_x2 else _p(x2, _dp)))
        return x1, (None if resident_x2 else x2[:2 * self._P])

    def last_solve_info(self):
        """bit 0: a 2x2 block was not positive definite (the solve raised EMBA_ERR_NUMERIC); bit 1: a pivot of S vanished (zero update)."""
        v = C.c_int32(0)
        self._check(self._L.emba_last_solve_info(self._ctx, C.byref(v)))
        return v.value

    # -- sharded Schur solve: primitives around the caller's two collectives (emba_amd.sharded.ShardedLEGM.solveNormalEq) --------
    def solve_shard_size(self):
        n = C.c_size_t(0)
        self._check(self._L.emba_solve_shard_size(self._ctx, C.byref(n)))
        return n.value

    def solve_shard_count(self, n_ranks):
        counts = np.zeros(n_ranks, dtype=np.uint64)
        self._check(self._L.emba_solve_shard_count(self._ctx, int(n_ranks), counts.ctypes.data_as(_lib._szp)))
        return counts.astype(np.int64)

    def solve_shard_pack(self, n_ranks, send_ptr):
        self._check(self._L.emba_solve_shard_pack(self._ctx, int(n_ranks), C.c_void_p(send_ptr)))

    def solve_shard_cached(self, rank, n_ranks):
        """n_recv if this rank still holds the records it received for the current equations (a re-solve may skip the exchange), else None."""
        f, n = C.c_int32(0), C.c_size_t(0)
        self._check(self._L.emba_solve_shard_cached(self._ctx, int(rank), int(n_ranks), C.byref(f), C.byref(n)))
        return int(n.value) if f.value else None

    # -- sharded solveNormalEqCG: the per-rank steps (emba_cg_shard_*), driven by emba_amd.sharded.ShardedLEGM.solveNormalEqCG
    def cg_shard_size(self):
        n = C.c_size_t(0)
        self._check(self._L.emba_cg_shard_size(self._ctx, C.byref(n)))
        return n.value

    def cg_shard_begin(self, rank, n_ranks, recv_ptr, n_recv, lam, fix_first_pose, red_ptr):
        self._check(self._L.emba_cg_shard_begin(self._ctx, int(rank), int(n_ranks), C.c_void_p(recv_ptr), int(n_recv), float(lam), 1 if fix_first_pose else 0, C.c_void_p(red_ptr)))

    def cg_shard_apply(self, red_ptr):
        self._check(self._L.emba_cg_shard_apply(self._ctx, C.c_void_p(red_ptr)))

    def cg_shard_pt(self, red_ptr):
        v = C.c_double(0)
        self._check(self._L.emba_cg_shard_pt(self._ctx, C.c_void_p(red_ptr), C.byref(v)))
        return v.value

    def cg_shard_update(self, alpha, red_ptr):
        self._check(self._L.emba_cg_shard_update(self._ctx, float(alpha), C.c_void_p(red_ptr)))

    def cg_shard_direction(self, beta):
        self._check(self._L.emba_cg_shard_direction(self._ctx, float(beta)))

    def cg_shard_end(self, x2_full_ptr):
        x1 = np.zeros(3 * self.K)
        self._check(self._L.emba_cg_shard_end(self._ctx, _p(x1, _dp), C.c_void_p(x2_full_ptr)))
        return x1

    def solve_shard_partial(self, rank, n_ranks, recv_ptr, n_recv, lam, S_ptr):
        self._check(self._L.emba_solve_shard_partial(self._ctx, int(rank), int(n_ranks), C.c_void_p(recv_ptr), int(n_recv), float(lam), C.c_void_p(S_ptr)))

    def solve_shard_finish(self, rank, n_ranks, recv_ptr, n_recv, lam, fix_first_pose, S_ptr, x2_ptr):
        x1 = np.zeros(3 * self.K)
        self._check(self._L.emba_solve_shard_finish(self._ctx, int(rank), int(n_ranks), C.c_void_p(recv_ptr), int(n_recv), float(lam),
                                                    1 if fix_first_pose else 0, C.c_void_p(S_ptr), _p(x1, _dp), C.c_void_p(x2_ptr)))
        return x1

    def solveNormalEqCG(self, lam, fix_first_pose=False, max_iter=100, tol=1e-6, resident_x2=False):
        """model.cpp:794-840 (Eigen ConjugateGradient, 100 iterations, tolerance 1e-6) on the device: returns (x1, x2, iterations, error);
        resident_x2 as in solveNormalEq."""
        self.last_counts()
        x1 = np.zeros(3 * self.K)
        x2 = None if resident_x2 else np.zeros(2 * max(self._P, 1))
        it = C.c_int32(0); err = C.c_double(0)
        self._check(self._L.emba_solve_normal_eq_cg(self._ctx, float(lam), 1 if fix_first_pose else 0, int(max_iter), float(tol), _p(x1, _dp),
                                                    None if resident_x2 else _p(x2, _dp), C.byref(it), C.byref(err)))
        return x1, (None if resident_x2 else x2[:2 * self._P]), it.value, err.value

    def updateMap(self, x2, damping_factor):
        """model.cpp:863-903 on the device-resident map: builds the TRIAL map (active += damping*x2, all other pixels 0) that
        the following evaluateDataError(traj, None, None) uses; report the LM decision with acceptMap() / rejectMap().
        x2: a host array; None = the x2 the last solveNormalEq[CG] left on the device (resident_x2=True there saves the download
        too); or an int = device address of 2P doubles (a sharded host's all-reduced x2)."""
        if x2 is None:
            self._check(self._L.emba_update_map(self._ctx, None, float(damping_factor)))
        elif isinstance(x2, int):
            self._check(self._L.emba_update_map_dev(self._ctx, C.c_void_p(x2), float(damping_factor)))
        else:
            x2 = np.ascontiguousarray(x2, dtype=np.float64)
            self._check(self._L.emba_update_map(self._ctx, _p(x2, _dp), float(damping_factor)))

    def acceptMap(self):
        self._check(self._L.emba_map_accept(self._ctx))

    def rejectMap(self):
        self._check(self._L.emba_map_reject(self._ctx))

    def rejectTrial(self):
        """emba_trial_reject: the last evaluation was a rejected trial; the equations formed before it are current again."""
        self._check(self._L.emba_trial_reject(self._ctx))

    def downloadMap(self):
        Gx = np.empty((self.H, self.W)); Gy = np.empty((self.H, self.W))
        self._check(self._L.emba_download_map(self._ctx, _p(Gx, _dp), _p(Gy, _dp)))
        return Gx, Gy

    def set_cost(self, cost_type="quadratic", a=0.0):
        """Declare the robust cost of the formNormalEqIRLS calls to come, so that evaluations accumulate the IRLS-weighted
        per-pixel sums directly (speed only; see emba_set_cost)."""
        self._check(self._L.emba_set_cost(self._ctx, COST_TYPES[cost_type], float(a)))

    def reconstructIntensity(self, Gx=None, Gy=None, download=True):
        """poisson_reconstruction::reconstructFromGradient (poisson_reconstruction.cpp:9-50; solver.cpp:417,471): the intensity
        panorama whose gradient is (Gx, Gy) — or the device-resident map when both are None."""
        if (Gx is None) != (Gy is None):
            raise ValueError("pass both Gx and Gy, or neither")
        if Gx is not None:
            Gx = np.ascontiguousarray(Gx, dtype=np.float64); Gy = np.ascontiguousarray(Gy, dtype=np.float64)
            if Gx.shape != (self.H, self.W) or Gy.shape != (self.H, self.W):
                raise ValueError("Gx/Gy must be pano_height x pano_width float64")
        M = np.empty((self.H, self.W)) if download else None
        self._check(self._L.emba_reconstruct_intensity(self._ctx, _p(Gx, _dp), _p(Gy, _dp), _p(M, _dp)))
        return M

    def A12_sparse(self):
        """Rank-1 factors of A12 (one per measurement candidate; pix == -1 marks outliers/inactive)."""
        _, M = self.event_counts()
        out = dict(cp_c=np.zeros(M, np.int32), cp_p=np.zeros(M, np.int32), pix=np.zeros(M, np.int32), w=np.zeros(M),
                   jc=np.zeros((M, 6)), jp=np.zeros((M, 6)), dp=np.zeros((M, 2)))
        self._check(self._L.emba_get_A12_sparse(self._ctx, _p(out["cp_c"], _i32p), _p(out["cp_p"], _i32p), _p(out["pix"], _i32p),
                                                _p(out["w"], _dp), _p(out["jc"], _dp), _p(out["jp"], _dp), _p(out["dp"], _dp)))
        return out

    def dump_state(self, fields=None):
        """Per-event State_LEGM (state.h:56-83) in original event order, for parity tests.  fields: subset of
        (pm, D, cp_idx, inlier_idx, pm_int, dp, Gpm, temp) — only those are produced (a 100 M-event pm dump is 1.6 GB, all of it 17 GB)."""
        n = self.n_events
        shapes = dict(pm=((n, 2), np.float64), D=((n, 2, 6), np.float64), cp_idx=((n,), np.int32), inlier_idx=((n,), np.int32),
                      pm_int=((n, 2), np.int32), dp=((n, 2), np.float64), Gpm=((n, 2), np.float64), temp=((n, 2), np.float64))
        want = list(shapes) if fields is None else list(fields)
        d = {k: np.zeros(shapes[k][0], shapes[k][1]) for k in want}
        g = lambda k, ty: _p(d[k], ty) if k in d else None
        self._check(self._L.emba_dump_state(self._ctx, g("pm", _dp), g("D", _dp), g("cp_idx", _i32p), g("inlier_idx", _i32p),
                                            g("pm_int", _i32p), g("dp", _dp), g("Gpm", _dp), g("temp", _dp)))
        return d

    # -- phase-level, HBM-resident interface (bench.py, sharded host) -------------------------------
    def upload_map(self, Gx, Gy):
        Gx = np.ascontiguousarray(Gx, dtype=np.float64); Gy = np.ascontiguousarray(Gy, dtype=np.float64)
        self._check(self._L.emba_upload_map(self._ctx, _p(Gx, _dp), _p(Gy, _dp)))
        self._check(self._L.emba_sync(self._ctx))

    def bind_map_dev(self, gx_ptr, gy_ptr):
        self._check(self._L.emba_bind_map_dev(self._ctx, C.c_void_p(gx_ptr), C.c_void_p(gy_ptr)))

    def bind_exchange_buffers(self, count_ptr, pack_ptr, pack_cap):
        self._check(self._L.emba_bind_exchange_buffers(self._ctx, C.c_void_p(count_ptr) if count_ptr else None,
                                                       C.c_void_p(pack_ptr) if pack_ptr else None, int(pack_cap)))

    def count_map_ready(self):
        """Make the device count map hold the counts of the last evaluation (call before reading / all-reducing a bound map)."""
        self._check(self._L.emba_count_map_ready(self._ctx))

    def count_compress(self, u8_ptr, cap):
        self._check(self._L.emba_count_compress(self._ctx, C.c_void_p(u8_ptr), int(cap)))

    def count_expand(self, u8_ptr):
        self._check(self._L.emba_count_expand(self._ctx, C.c_void_p(u8_ptr)))

    def eval_launch(self, traj):
        knots = np.ascontiguousarray(traj.knots_xyzw, dtype=np.float64).reshape(-1, 4)
        self.K = knots.shape[0]
        self._keep = [knots]
        self._check(self._L.emba_eval_launch(self._ctx, _p(knots, _dp), self.K, int(traj.t0_ns), int(traj.dt_ns)))

    def eval_finish(self, want_ep=False, want_map=False, sync=True):
        if not sync and not want_ep and not want_map:      # enqueue only; counts via last_counts() after a sync point
            self._check(self._L.emba_eval_finish(self._ctx, None, None, None))
            return None, None, None
        n_inl = C.c_size_t(0)
        ep = np.empty(max(self.n_events, 1)) if want_ep else None
        nem = np.empty((self.H, self.W), dtype=np.int32) if want_map else None
        self._check(self._L.emba_eval_finish(self._ctx, _p(ep, _dp), C.byref(n_inl), _p(nem, _i32p)))
        return n_inl.value, (ep[:n_inl.value] if want_ep else None), nem

    def form_active(self, thres, sync=True):
        if not sync:
            self._check(self._L.emba_form_active(self._ctx, int(thres), None, None))
            return None, None
        P, pl = C.c_size_t(0), C.c_size_t(0)
        self._check(self._L.emba_form_active(self._ctx, int(thres), C.byref(P), C.byref(pl)))
        self._P = P.value
        return P.value, pl.value

    def step_form_active(self, thres, global_u8_ptr=None):
        """F1 of a resident step on a rank of a sharded window (emba_step_form_active): activity from the all-reduced saturated byte counts."""
        self._check(self._L.emba_step_form_active(self._ctx, int(thres), C.c_void_p(global_u8_ptr) if global_u8_ptr else None))

    def form_accumulate(self, cost_type="quadratic", a=0.0):
        self._check(self._L.emba_form_accumulate(self._ctx, None, COST_TYPES[cost_type], float(a)))

    def form_finish(self, alpha, download=False, dense_A12=False):
        if download:
            self.last_counts()
            return self._finish(alpha, dense_A12)
        self._check(self._L.emba_form_finish(self._ctx, float(alpha), None, None, None, 0, None, None, None))
        return None

    def last_counts(self):
        a, b = C.c_size_t(0), C.c_size_t(0)
        self._check(self._L.emba_last_counts(self._ctx, C.byref(a), C.byref(b)))
        self._P = b.value
        return a.value, b.value

    def step(self, traj, thres_valid_pixel, alpha, cost_type="quadratic", a=0.0):
        """One whole resident step (evaluateDataError + formNormalEq[IRLS] + applyL2Reg) in a single library call."""
        if getattr(self, "_step_traj", None) is not traj:       # cache the contiguous knot array of this trajectory object
            self._step_knots = np.ascontiguousarray(traj.knots_xyzw, dtype=np.float64).reshape(-1, 4)
            self._step_traj = traj
        knots = self._step_knots
        self.K = knots.shape[0]
        a_, b_ = C.c_size_t(0), C.c_size_t(0)
        self._check(self._L.emba_step(self._ctx, _p(knots, _dp), self.K, int(traj.t0_ns), int(traj.dt_ns), int(thres_valid_pixel),
                                      COST_TYPES[cost_type], float(a), float(alpha), C.byref(a_), C.byref(b_)))
        self._P = b_.value
        return a_.value, b_.value

    def get_ep(self):
        """The last evaluation's residual vector in the reference's order, as the resident step left it on the device (or compacted now)."""
        n = C.c_size_t(0)
        ep = np.empty(max(int(self.n_events), 1), dtype=np.float64)
        self._check(self._L.emba_get_ep(self._ctx, _p(ep, _dp), ep.size, C.byref(n)))
        return ep[: n.value].copy()

    def set_option(self, name, value):
        """Tuning / A-B switch by name (include/emba_hip.h: emba_set_option); results do not depend on any of them."""
        self._check(self._L.emba_set_option(self._ctx, name.encode(), int(value)))

    def get_option(self, name):
        v = C.c_int32(0)
        self._check(self._L.emba_get_option(self._ctx, name.encode(), C.byref(v)))
        return v.value

    def compact_ep(self):
        """Enqueue the residual compaction into the reference-order `ep` vector on the device (emba_compact_ep)."""
        self._check(self._L.emba_compact_ep(self._ctx))

    def sync(self):
        self._check(self._L.emba_sync(self._ctx))

    def timer_start(self, slot=0):
        self._check(self._L.emba_timer_start(self._ctx, slot))

    def timer_stop(self, slot=0):
        self._check(self._L.emba_timer_stop(self._ctx, slot))

    def timer_ms(self, slot=0):
        ms = C.c_float(0)
        self._check(self._L.emba_timer_elapsed_ms(self._ctx, slot, C.byref(ms)))
        return ms.value

    def enable_kernel_timing(self, on=True, slot=0):
        """on: the next step's warp / Gram kernels are bracketed by HIP events in `slot` (< 16); read them with last_kernel_ms() at once or with
        kernel_ms_slot(slot) after the loop (no host wait inside it)."""
        self._check(self._L.emba_enable_kernel_timing(self._ctx, (1 + int(slot)) if on else 0))

    def kernel_ms_slot(self, slot):
        a, b = C.c_float(0), C.c_float(0)
        self._check(self._L.emba_kernel_ms_slot(self._ctx, int(slot), C.byref(a), C.byref(b)))
        return a.value, b.value

    def last_kernel_ms(self):
        a, b = C.c_float(0), C.c_float(0)
        self._check(self._L.emba_last_kernel_ms(self._ctx, C.byref(a), C.byref(b)))
        return a.value, b.value

    def kernel_timing_all(self, on=True):
        """Sampled steps also record an event in front of their first launch: kernel_ms_all(slot) then returns the four intervals that tile the
        step's device time (prep || pose || texel, warp, post-warp launch A, Gram)."""
        self._check(self._L.emba_kernel_timing_all(self._ctx, 1 if on else 0))

    def kernel_ms_all(self, slot):
        v = (C.c_float * 4)()
        self._check(self._L.emba_kernel_ms_all(self._ctx, int(slot), v))
        return [float(x) for x in v]

    def bracket_overhead_us(self, reps=50):
        """Mean HIP-event interval around an EMPTY kernel queued behind running work: what a bracket reads for a kernel of zero length."""
        us = C.c_float(0)
        self._check(self._L.emba_bracket_overhead_us(self._ctx, int(reps), C.byref(us)))
        return us.value

    def clock_probe(self):
        """Device clock attributes and the shader clock measured by a probe kernel while the whole chip is busy (emba_clock_probe)."""
        sclk, us, ratio = C.c_double(0), C.c_double(0), C.c_double(0)
        cr, mr, wr = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        self._check(self._L.emba_clock_probe(self._ctx, C.byref(sclk), C.byref(us), C.byref(ratio), C.byref(cr), C.byref(mr), C.byref(wr)))
        return {"sclk_mhz_measured": sclk.value, "probe_us": us.value, "cycles_per_dependent_add": ratio.value,
                "attr_clock_rate_khz": cr.value, "attr_mem_clock_rate_khz": mr.value, "attr_wall_clock_rate_khz": wr.value}

    def pci_bus_id(self):
        buf = C.create_string_buffer(32)
        self._check(self._L.emba_device_pci_bus_id(self._ctx, buf, 32))
        return buf.value.decode()
