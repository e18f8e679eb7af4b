"""TEST INFRASTRUCTURE: a CPU 'engine' with the phase interface ShardedLEGM expects, built from the oracle's leaf
functions (spline_eval, warp, hessian) on the events a rank is GIVEN (local + halo) and nothing else.  It lets the
world_size-2 gloo tests exercise the real sharding logic (emba_amd/sharded.py: batch-aligned ranges, per-pixel halo,
count all-reduce, pack all-reduce, L2 after the reduce) without a GPU.  The arithmetic follows model.cpp like the oracle
does; sizes are kept small (pure-Python loops)."""
import numpy as np
import torch

from oracle import oracle as O


class OracleShardEngine:
    def __init__(self, w):
        self.w = w
        self.o = O.OracleLEGM(w.sensor_w, w.sensor_h, w.pano_w, w.pano_h, w.lut, w.C_th)

    def bind_exchange(self, count_tensor, pack_tensor):
        self.count, self.pack = count_tensor, pack_tensor

    def declare_cost(self, cost_type, a):
        """like the device engine after emba_set_cost: form_active leaves the final A22 | b2 rows in the pack, form_accumulate only
        writes the A11 | b1 head — ShardedLEGM may then reduce the rows while the head is still being formed"""
        self.cost = (cost_type, a)
        return True

    def _weight(self, e):
        cost_type, a = getattr(self, "cost", ("quadratic", 0.0))
        if cost_type == "cauchy":
            return 1.0 / (1.0 + a * e * e)
        if cost_type == "huber":
            return 1.0 if abs(e) < a else a / abs(e)
        return 1.0

    def set_events(self, events, halo):
        self.ev, self.halo = events, halo

    def upload_map(self, Gx, Gy):
        self.Gx, self.Gy = np.asarray(Gx), np.asarray(Gy)

    def eval_launch(self, traj):
        w, ev = self.w, self.ev
        self.K = traj.size()
        n_used = (ev.size() // 100) * 100
        Gxx, Gxy, Gyy = O.hessian(self.Gx, self.Gy)
        items = []   # (pixel, order, pol, pose_time, is_halo)
        hx, hy, ht = self.halo
        for i in range(len(hx)):
            items.append((int(hy[i]) * w.sensor_w + int(hx[i]), -1 - (len(hx) - i), 0, int(ht[i]), True, int(hx[i]), int(hy[i])))
        for k in range(n_used):
            b = k // 100
            tb = O.batch_mid_ns(ev.t_ns[100 * b], ev.t_ns[100 * b + 99])
            items.append((int(ev.y[k]) * w.sensor_w + int(ev.x[k]), k, int(ev.polarity[k]), tb, False, int(ev.x[k]), int(ev.y[k])))
        items.sort(key=lambda it: (it[0], it[1]))
        pose_cache = {}

        def state(it):
            t = it[3]
            if t not in pose_cache:
                pose_cache[t] = O.spline_eval(traj.knots_xyzw, traj.t0_ns, traj.dt_ns, t)
            q, R, s, J36 = pose_cache[t]
            pm, J23 = self.o.warp(it[5], it[6], q)
            return pm, J23 @ J36, s

        count = np.zeros(w.pano_h * w.pano_w, dtype=np.int32)
        self.meas = []
        prev = None
        for it in items:
            cur = (it, state(it))
            if prev is not None and prev[0][0] == it[0] and not it[4]:
                (pm, D, s), (pmp, Dp, sp) = cur[1], prev[1]
                dp = pm - pmp
                if np.hypot(dp[0], dp[1]) <= 10:
                    rx, ry = np.floor(pm[0] + 0.5), np.floor(pm[1] + 0.5)
                    if 0 <= rx < w.pano_w and 0 <= ry < w.pano_h:
                        pi = int(ry) * w.pano_w + int(rx)
                        g = np.array([self.Gx.ravel()[pi], self.Gy.ravel()[pi]])
                        e = 2 * (it[2] - 0.5) * w.C_th - g @ dp
                        G2 = np.array([[Gxx.ravel()[pi], Gxy.ravel()[pi]], [Gxy.ravel()[pi], Gyy.ravel()[pi]]])
                        temp = g + dp @ G2
                        count[pi] += 1
                        self.meas.append(dict(pix=it[0], k=it[1], pi=pi, e=e, dp=dp, jc=temp @ D, jp=-g @ Dp, c=s, p=sp))
            prev = cur
        self.count.copy_(torch.from_numpy(count))

    def count_compress(self, u8, cap):
        u8.copy_(torch.clamp(self.count, max=cap).to(torch.uint8))

    def count_expand(self, u8):
        self.count.copy_(u8.to(torch.int32))

    def eval_finish(self):
        self.ep = np.array([m["e"] for m in self.meas])          # already pixel-major, then time
        self.ep_pix = np.array([m["pix"] for m in self.meas], dtype=np.int64)

    def form_active(self, thres, sync=True):
        self._recv_cache = None             # new equations: the records an owner received for the previous ones are stale (emba_solve_shard_cached)
        self.meas_form = self.meas          # the measurements these equations consist of (a later, rejected, trial evaluation must not replace them:
                                            # the device keeps them in its second record set, the reference in its host copies of A and b)
        cnt = self.count.numpy()
        self.active = np.nonzero(cnt >= thres)[0]
        self.compact = -np.ones(cnt.size, dtype=np.int64)
        self.compact[self.active] = np.arange(self.active.size)
        self.P = self.active.size
        K, P = self.K, self.P
        self.pack_len = 9 * K * K + 3 * K + 5 * P
        A22b2 = np.zeros((max(P, 1), 5))
        for m in self.meas:
            ci = self.compact[m["pi"]]
            if ci < 0:
                continue
            e, dp = m["e"], m["dp"]
            wgt = self._weight(e)
            A22b2[ci] += [wgt * dp[0] * dp[0], wgt * dp[0] * dp[1], wgt * dp[1] * dp[1], dp[0] * wgt * e, dp[1] * wgt * e]
        self.pack[9 * K * K + 3 * K: self.pack_len].copy_(torch.from_numpy(A22b2[:P].ravel()))
        return self.P, self.pack_len

    def form_accumulate(self, cost_type, a):
        norm = lambda c: (c[0], c[1] if c[0] != "quadratic" else 0.0)     # (the parameter means nothing for the quadratic cost)
        assert norm((cost_type, a)) == norm(getattr(self, "cost", (cost_type, a))), "the stand-in forms A22 | b2 with the DECLARED cost"
        self.cost = (cost_type, a)
        K = self.K
        A11 = np.zeros((3 * K, 3 * K)); b1 = np.zeros(3 * K)
        for m in self.meas:
            if self.compact[m["pi"]] < 0:
                continue
            e = m["e"]
            wgt = self._weight(e)
            v = np.concatenate([m["jc"], m["jp"]])
            idx = np.r_[3 * m["c"]:3 * m["c"] + 6, 3 * m["p"]:3 * m["p"] + 6]
            np.add.at(A11, (idx[:, None], idx[None, :]), wgt * np.outer(v, v))
            np.add.at(b1, idx, v * wgt * e)
        # ONLY the head: the A22 | b2 rows may be in an all-reduce right now
        self.pack[: 9 * K * K + 3 * K].copy_(torch.from_numpy(np.concatenate([A11.ravel(order="F"), b1])))

    def form_finish(self, alpha, download):
        K, P = self.K, self.P
        pk = self.pack[: self.pack_len].numpy().copy()
        A22b2 = pk[9 * K * K + 3 * K:].reshape(P, 5)
        if alpha:
            A22b2[:, 0] += alpha; A22b2[:, 2] += alpha
            A22b2[:, 3] -= alpha * self.Gx.ravel()[self.active]
            A22b2[:, 4] -= alpha * self.Gy.ravel()[self.active]
        self.pack[: self.pack_len].copy_(torch.from_numpy(pk))    # in place, like the device engine: the solve phases read the regularised pack
        A22 = np.stack([A22b2[:, 0], A22b2[:, 1], A22b2[:, 1], A22b2[:, 2]], axis=1).reshape(P, 2, 2)
        return dict(A11=pk[:9 * K * K].reshape(3 * K, 3 * K, order="F"), b1=pk[9 * K * K:9 * K * K + 3 * K], active=self.active,
                    A22=A22, b2=A22b2[:, 3:5].ravel(), P=P)

    def last_counts(self):
        return len(self.meas), self.P

    def sync(self):
        pass

    # ---- sharded Schur solve (ShardedLEGM.solveNormalEq): the phase interface of emba_solve_shard_* on numpy --------------------
    # A packed record is 16 doubles like the device's: jc[6] jp[6] e {compact pixel, pair} dp[2]; here {compact pixel, pair} travel as ONE
    # exactly representable double (pixel + 2^24 (4096 c + p)) — the layout is private to an engine, only its size is shared.
    @staticmethod
    def _owner(k, P, n):
        r = min((k * n) // max(P, 1), n - 1)
        while r + 1 < n and (P * (r + 1)) // n <= k:
            r += 1
        while r > 0 and (P * r) // n > k:
            r -= 1
        return r

    def _active_records(self):
        return [(m, int(self.compact[m["pi"]])) for m in self.meas_form if self.compact[m["pi"]] >= 0]

    def solve_shard_size(self):
        return (3 * self.K + 1) ** 2

    def solve_shard_count(self, n_ranks):
        cnt = np.zeros(n_ranks, dtype=np.int64)
        for _, ci in self._active_records():
            cnt[self._owner(ci, self.P, n_ranks)] += 1
        return cnt

    def solve_shard_pack(self, n_ranks, send):
        recs = sorted(self._active_records(), key=lambda mc: self._owner(mc[1], self.P, n_ranks))     # stable: grouped by owner
        out = np.zeros((max(len(recs), 1), 16))
        for i, (m, ci) in enumerate(recs):
            out[i, :6] = m["jc"]; out[i, 6:12] = m["jp"]; out[i, 12] = m["e"]
            out[i, 13] = float(ci) + float(1 << 24) * (m["c"] * 4096 + m["p"])
            out[i, 14] = m["dp"][0]; out[i, 15] = m["dp"][1]
        send[: out.size].copy_(torch.from_numpy(out.ravel()))

    def solve_shard_cached(self, rank, n_ranks):
        """like the device: n_recv while this owner still holds the records it received for the current equations, else None"""
        c = getattr(self, "_recv_cache", None)
        return c.shape[0] if c is not None else None

    def _received(self, recv, n_recv):
        if recv is None:
            assert self._recv_cache is not None and self._recv_cache.shape[0] == n_recv, "no cached records for these equations"
            return self._recv_cache
        self._recv_cache = recv[: n_recv * 16].numpy().reshape(n_recv, 16).copy()
        return self._recv_cache

    def _pixel_terms(self, recv, n_recv, lam):
        """per OWNED pixel: A12 columns (3K x 2) from the received records, C = A22m, b2 (global, from the reduced pack)."""
        K = self.K
        pk = self.pack[: self.pack_len].numpy()
        A22b2 = pk[9 * K * K + 3 * K:].reshape(self.P, 5)
        R = self._received(recv, n_recv)
        cols = {}
        for r in R:
            code = int(r[13]); ci = code % (1 << 24); pair = code >> 24; c, p = pair // 4096, pair % 4096
            v = np.concatenate([r[:6], r[6:12]]); dp = r[14:16]
            idx = np.r_[3 * c:3 * c + 6, 3 * p:3 * p + 6]
            A = cols.setdefault(ci, np.zeros((3 * K, 2)))
            np.add.at(A, (idx[:, None], np.arange(2)[None, :]), self._weight(r[12]) * np.outer(v, dp))   # model.cpp:483-487 / 679-683 (IRLS weight)
        out = {}
        for ci, A in cols.items():
            a = A22b2[ci]
            C = np.array([[a[0] * (1 + lam), a[1]], [a[1], a[2] * (1 + lam)]])            # A22m = A22 + lambda diag(A22), :743-759
            out[ci] = (A, C, a[3:5])
        return out

    def solve_shard_partial(self, rank, n_ranks, recv, n_recv, lam, S):
        n = 3 * self.K
        Sa = np.zeros((n + 1, n + 1))
        for ci, (A, C, b2) in self._pixel_terms(recv, n_recv, lam).items():
            W = A @ np.linalg.inv(C)
            Sa[:n, :n] -= W @ A.T
            Sa[:n, n] -= W @ b2
        S.copy_(torch.from_numpy(Sa.ravel()))

    def solve_shard_finish(self, rank, n_ranks, recv, n_recv, lam, fix_first_pose, S, x2):
        K = self.K; n = 3 * K
        pk = self.pack[: self.pack_len].numpy()
        A11 = pk[:9 * K * K].reshape(n, n, order="F"); b1 = pk[9 * K * K:9 * K * K + n]
        Sa = S.numpy().reshape(n + 1, n + 1)
        Sm = Sa[:n, :n] + A11 + lam * np.diag(np.diag(A11))                               # A11m, :728-730
        rhs = Sa[:n, n] + b1
        sk = 3 if fix_first_pose else 0
        x1 = np.zeros(n)
        x1[sk:] = O.ldlt_solve(Sm[sk:, sk:], rhs[sk:])[0]                                  # S.ldlt().solve(...), :789 (Eigen's pivoted LDLT restated)
        xx = np.zeros(2 * max(self.P, 1))
        for ci, (A, C, b2) in self._pixel_terms(recv, n_recv, lam).items():
            xx[2 * ci:2 * ci + 2] = np.linalg.solve(C, b2 - A.T @ x1)                    # :790-791
        x2[: xx.size].copy_(torch.from_numpy(xx))            # (the caller's buffer carries a status word behind the 2P entries)
        return x1

    # ---- sharded solveNormalEqCG: the per-rank steps (the device's emba_cg_shard_*), dense per-pixel blocks on the CPU -------------------
    def cg_shard_size(self):
        return 3 * self.K + 2

    def cg_shard_begin(self, rank, n_ranks, recv, n_recv, lam, fix_first_pose, red):
        K = self.K; n = 3 * K
        pk = self.pack[: self.pack_len].numpy()
        A11 = pk[:9 * K * K].reshape(n, n, order="F"); b1 = pk[9 * K * K:9 * K * K + n]
        sk = 3 if fix_first_pose else 0
        terms = self._pixel_terms(recv, n_recv, lam)            # ci -> (A12 columns 3K x 2, C = A22m, b2)
        own = sorted(terms)
        lo, hi = (self.P * rank) // n_ranks, (self.P * (rank + 1)) // n_ranks
        pix = list(range(lo, hi))
        A22b2 = pk[9 * K * K + 3 * K:].reshape(self.P, 5)
        A11m = A11 + lam * np.diag(np.diag(A11))
        A11m[:sk, :] = 0; A11m[:, :sk] = 0
        cols = {}
        for ci in pix:
            a = A22b2[ci]
            C = np.array([[a[0] * (1 + lam), a[1]], [a[1], a[2] * (1 + lam)]])
            A = terms[ci][0].copy() if ci in terms else np.zeros((n, 2))
            A[:sk, :] = 0
            cols[ci] = (A, C, a[3:5].copy())
        assert set(own) <= set(pix)
        d1 = np.diag(A11) * (1 + lam)
        bb1 = b1.copy(); bb1[:sk] = 0
        inv = lambda d: np.where(d != 0, 1.0 / np.where(d != 0, d, 1.0), 1.0)
        x = np.zeros(n + 2 * len(pix)); r = np.concatenate([bb1] + [cols[ci][2] for ci in pix]) if pix else bb1.copy()
        invd = np.concatenate([inv(d1)] + [inv(np.array([cols[ci][1][0, 0], cols[ci][1][1, 1]])) for ci in pix]) if pix else inv(d1)
        p = invd * r
        self._cg = dict(rank=rank, n=n, pix=pix, cols=cols, A11m=A11m, x=x, r=r, p=p, invd=invd, z=None, t=None)
        frm = 0 if rank == 0 else n
        out = np.zeros(n + 2)
        out[n] = float(np.dot(r[frm:], r[frm:])); out[n + 1] = float(np.dot(r[frm:], p[frm:]))
        red.copy_(torch.from_numpy(out))

    def cg_shard_apply(self, red):
        g = self._cg; n = g["n"]; p = g["p"]
        t = np.zeros_like(p)
        if g["rank"] == 0:
            t[:n] = g["A11m"] @ p[:n]
        for k, ci in enumerate(g["pix"]):
            A, C, _ = g["cols"][ci]
            v2 = p[n + 2 * k:n + 2 * k + 2]
            t[:n] += A @ v2
            t[n + 2 * k:n + 2 * k + 2] = A.T @ p[:n] + C @ v2
        g["t"] = t
        out = np.zeros(n + 2)
        out[:n] = t[:n]; out[n] = float(np.dot(p[n:], t[n:]))
        red.copy_(torch.from_numpy(out))

    def cg_shard_pt(self, red):
        g = self._cg; n = g["n"]
        rr = red.numpy()
        g["t"][:n] = rr[:n]
        return float(np.dot(g["p"][:n], g["t"][:n]) + rr[n])

    def cg_shard_update(self, alpha, red):
        g = self._cg; n = g["n"]
        g["x"] = g["x"] + alpha * g["p"]
        g["r"] = g["r"] - alpha * g["t"]
        g["z"] = g["invd"] * g["r"]
        frm = 0 if g["rank"] == 0 else n
        red[n] = float(np.dot(g["r"][frm:], g["r"][frm:])); red[n + 1] = float(np.dot(g["r"][frm:], g["z"][frm:]))

    def cg_shard_direction(self, beta):
        g = self._cg
        g["p"] = g["z"] + beta * g["p"]

    def cg_shard_end(self, x2):
        g = self._cg; n = g["n"]
        xx = np.zeros(2 * max(self.P, 1))
        for k, ci in enumerate(g["pix"]):
            xx[2 * ci:2 * ci + 2] = g["x"][n + 2 * k:n + 2 * k + 2]
        x2[: xx.size].copy_(torch.from_numpy(xx))
        return g["x"][:n].copy()

    # ---- what ShardedModel asks of "this rank's LEGM" in the LM loop (emba_amd.solver.solve_time_window over ranks) -------------
    @property
    def H(self):
        return self.w.pano_h

    @property
    def W(self):
        return self.w.pano_w

    def set_cost(self, cost_type="quadratic", a=0.0):
        self.cost = (cost_type, a)

    def dataCost(self, cost_type="quadratic", a=0.0):
        """this rank's measurements only (solver.cpp:88 / model.cpp:279-314)"""
        e = np.array([m["e"] for m in self.meas])
        return O.data_cost(e, {"quadratic": 0, "huber": 1, "cauchy": 2}[cost_type], a)

    def regCost(self, alpha):
        return O.reg_cost(self.Gx, self.Gy, alpha)

    def updateMap(self, x2, damping):
        self._cur = (self.Gx, self.Gy)
        self.Gx, self.Gy = O.update_map(self.active.astype(np.uint32), np.asarray(x2), damping, self.Gx, self.Gy)

    def acceptMap(self):
        self._cur = None

    def rejectMap(self):
        self.Gx, self.Gy = self._cur
        self._cur = None

    def downloadMap(self):
        return self.Gx, self.Gy
