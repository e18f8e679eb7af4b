"""Multi-GPU host for the hot path: one process per GPU, events sharded by time, two all-reduces per
Gauss-Newton iteration over RCCL/xGMI (torch.distributed backend "nccl") — SURVEY.md §8e.

The reference is single-process; this is new work, not a port.  The protocol per iteration:

    E1  every rank: pose table, texel pack, warp+residual+count+records on its own events   (no communication)
    X1  all_reduce(SUM) of the pixel-count map           — activity (count >= thres) is a global property (model.cpp:333,409);
        sent as one saturated BYTE per pixel (min(count, 255 // world)) when thres allows: the merged map only feeds `>= thres`
    E2  residual compaction (local), F1 active set from the GLOBAL counts (identical on every rank)
    F2  every rank accumulates its measurements into the pack [A11 | b1 | A22b2(P)]
    X2  all_reduce(SUM) of the fp64 pack (one call; A11 is 9K^2 doubles, A22b2 5P doubles)
    F3  applyL2Reg once, on the reduced pack (adding alpha*I on every rank before X2 would count it world_size times)

The sparse A12 factors stay sharded (each rank keeps the records of its own measurements).

Sharding rule: contiguous ranges of WHOLE 100-event batches cut on the GLOBAL batch grid (quirks Q1/Q3: a batch's pose is
the midpoint of its first and last timestamp), plus a halo: for every sensor pixel, the last event before the rank's range
(with the midpoint time of the batch it belongs to) so that the first local event at that pixel finds its predecessor.

`engine` is any object with the phase interface of emba_amd.LEGM (set_events, upload_map, bind_exchange, eval_launch,
eval_finish, form_active, form_accumulate, form_finish); the product passes emba_amd.LEGM.
"""
import math

import numpy as np

from ._lib import ERR_NUMERIC as _ERR_NUMERIC, EmbaError
from .legm import EventPacket

BATCH = 100  # model.cpp:78


def batch_mid_ns(t_first, t_last):
    """ros::Time/Duration midpoint of a batch (model.cpp:116-119), same integer/double steps as the library."""
    d = int(t_last) - int(t_first)
    dsec, dnsec = divmod(d, 1_000_000_000)
    half = (float(dsec) + 1e-9 * float(dnsec)) * 0.5
    hsec = math.floor(half)
    frac = (half - float(hsec)) * 1e9
    hnsec = int(math.floor(frac + 0.5)) if frac >= 0 else -int(math.floor(-frac + 0.5))   # round half away from zero
    hsec += hnsec // 1_000_000_000
    hnsec = hnsec % 1_000_000_000
    return int(t_first) + hsec * 1_000_000_000 + hnsec


def batch_ranges(n_events, world_size):
    """[(first_event, end_event)] per rank: whole batches, as even as possible, tail (n % 100) dropped (Q1)."""
    nb = n_events // BATCH
    base, rem = divmod(nb, world_size)
    out, b = [], 0
    for r in range(world_size):
        cnt = base + (1 if r < rem else 0)
        out.append((b * BATCH, (b + cnt) * BATCH))
        b += cnt
    return out


def shard_events(events, sensor_w, rank, world_size):
    """Returns (local EventPacket, halo=(x, y, batch_t_ns)) for `rank`."""
    n = events.size()
    lo, hi = batch_ranges(n, world_size)[rank]
    local = EventPacket(events.x[lo:hi], events.y[lo:hi], events.polarity[lo:hi], events.t_ns[lo:hi])
    if lo == 0:
        return local, (np.zeros(0, np.uint16), np.zeros(0, np.uint16), np.zeros(0, np.int64))
    pix = events.y[:lo].astype(np.int64) * sensor_w + events.x[:lo]
    # last occurrence of every pixel before `lo`: first occurrence in the reversed array
    rev = pix[::-1]
    _, first_rev = np.unique(rev, return_index=True)
    idx = np.sort(lo - 1 - first_rev)                       # global indices, ascending (time order)
    bt = np.array([batch_mid_ns(events.t_ns[(k // BATCH) * BATCH], events.t_ns[(k // BATCH) * BATCH + BATCH - 1]) for k in idx],
                  dtype=np.int64)
    return local, (events.x[idx].astype(np.uint16), events.y[idx].astype(np.uint16), bt)


def merge_ep(ep_parts, pix_parts):
    """Global residual vector in the reference's order (sensor pixel major, then time) from per-rank vectors.
    pix_parts[r][i] = sensor pixel of the event that produced ep_parts[r][i].  Ranks are time-ordered, so within
    a pixel rank r's measurements precede rank r+1's."""
    ep = np.concatenate(ep_parts)
    pix = np.concatenate(pix_parts)
    rank = np.concatenate([np.full(p.size, r) for r, p in enumerate(pix_parts)])
    pos = np.concatenate([np.arange(p.size) for p in pix_parts])
    order = np.lexsort((pos, rank, pix))
    return ep[order]


def _fill_done(dev):
    """A tensor torch has just zero-filled is about to be WRITTEN by the engine's kernels.  In production the engine runs on torch's current stream (HipEngine checks
    it) and the fill is ordered in front of them; where it runs on a stream of its own (the rank-thread tests: check_stream=False) nothing orders torch's fill kernel
    against the engine's first writes — with eight ranks saturating one device the fill was seen to run AFTER them once (round 6: the Schur sums of a rank zeroed, a
    wrong x1).  Draining torch's stream here costs microseconds in either case."""
    if dev.type == "cuda":
        import torch
        torch.cuda.current_stream(dev).synchronize()


def _device_sync(dev):
    """(the CPU stand-in engine of the gloo tests keeps its tensors on the host)"""
    if dev.type == "cuda":
        import torch
        torch.cuda.synchronize(dev)


class ShardedLEGM:
    """LEGM over `dist` (a torch.distributed-like module: all_reduce, get_rank, get_world_size)."""

    def __init__(self, engine, dist, count_tensor, pack_tensor, sensor_w, count_u8_tensor=None):
        """count_u8_tensor: optional uint8 tensor (one byte per panorama pixel) enabling the compressed exchange 1."""
        self.engine, self.dist = engine, dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.count, self.pack = count_tensor, pack_tensor
        self.count_u8 = count_u8_tensor
        self.sensor_w = sensor_w
        engine.bind_exchange(count_tensor, pack_tensor)
        self.P = 0
        self.pack_len = 0
        self.force_collectives = False   # rehearse the exchanges with a single rank

    def set_events(self, events):
        local, halo = shard_events(events, self.sensor_w, self.rank, self.world)
        self.engine.set_events(local, halo)
        self.n_local = local.size()
        # rank-invariant size of the largest shard: decisions that change the SEQUENCE of collectives (the split of exchange 2) must not
        # depend on this rank's own event count — shards differ by one batch, and ranks on different sides of a threshold would issue
        # different collectives
        self.n_max = max(hi - lo for lo, hi in batch_ranges(events.size(), self.world))
        return local

    def iteration(self, traj, thres_valid_pixel, alpha, cost_type="quadratic", a=0.0, download=False):
        """One evaluateDataError + formNormalEq[IRLS] + applyL2Reg over all ranks.  Map must be resident (upload_map)."""
        e, dist = self.engine, self.dist
        multi = self.world > 1 or self.force_collectives
        if not multi and not download and hasattr(e, "step"):     # single GPU, nothing to exchange: one library call per step
            n_inl, self.P = e.step(traj, thres_valid_pixel, alpha, cost_type, a)
            return n_inl, None
        self.evaluate(traj, cost_type, a)
        return self.form(thres_valid_pixel, alpha, cost_type, a, download)

    def evaluate(self, traj, cost_type="quadratic", a=0.0):
        """E1: evaluateDataError of this rank's shard (enqueue only).  Nothing is exchanged: the data cost is a sum over ranks
        (ShardedModel.dataCost), the count map is reduced when formNormalEq asks for it (form)."""
        e = self.engine
        multi = self.world > 1 or self.force_collectives
        # declare the cost before the evaluation: the per-pixel sums then already carry its weights and A22 | b2 are final after F1
        self._declared = bool(multi and hasattr(e, "declare_cost") and e.declare_cost(cost_type, a))
        e.eval_launch(traj)                                   # E1

    def form(self, thres_valid_pixel, alpha, cost_type="quadratic", a=0.0, download=False):
        """X1, E2, F1, F2, X2, F3 on the state of the last evaluate(): formNormalEq[IRLS] + applyL2Reg over all ranks."""
        e, dist = self.engine, self.dist
        multi = self.world > 1 or self.force_collectives
        # ... the split of exchange 2 pays once the Gram kernel is long enough to hide a collective behind (the head then costs a third collective's latency)
        split_x2 = getattr(self, "_declared", False) and (not hasattr(e, "x2_split_pays") or e.x2_split_pays(getattr(self, "n_max", 0)))
        cap = 255 // max(self.world, 1)
        # Round 5: a rank's step = the one-GPU step.  Where exchange 1 travels as saturated bytes, the per-pixel sums carry this cost's weights
        # (declared before the evaluation) and exchange 2 is not split (small shards: the rows are written inside the Gram launch), the engine forms in the
        # resident step's way on the exchanged BYTES (emba_step_form_active): launch A with active lists + zeroing, gather inside the Gram kernel, no clearing
        # pass in the next evaluation, no expansion into the int32 map.  Every condition is rank-invariant; the collectives are the same two as below.
        if (multi and getattr(self, "_declared", False) and not split_x2 and self.count_u8 is not None and 1 <= thres_valid_pixel <= cap
                and hasattr(e, "step_form_active") and getattr(e, "step_fast", True) and getattr(self, "n_max", 0) < 3_000_000):
            self.last_form_resident = True
            e.count_compress(self.count_u8, cap)                  # markers -> this rank's counts and their saturated bytes, one sweep
            dist.all_reduce(self.count_u8)                        # X1
            e.step_form_active(thres_valid_pixel, self.count_u8)  # E2, F1 (enqueue only)
            e.form_accumulate(cost_type, a)                       # F2: Gram with the active-set write + A22 | b2 gather inside
            n_inl, self.P = e.last_counts()                       # (polled: the gather's first block publishes P)
            self.pack_len = 9 * int(e.K) ** 2 + 3 * int(e.K) + 5 * self.P
            dist.all_reduce(self.pack[: self.pack_len])           # X2
            out = e.form_finish(alpha, download)                  # F3: applyL2Reg once, after the reduce
            n_inl, self.P = e.last_counts()
            return n_inl, out
        self.last_form_resident = False
        if multi:
            if self.count_u8 is not None and thres_valid_pixel <= cap and hasattr(e, "count_compress"):
                e.count_compress(self.count_u8, cap)          # int32 -> min(count, cap) bytes
                dist.all_reduce(self.count_u8)                # X1 (SUM) on a quarter of the bytes; cannot overflow: world * cap <= 255
                e.count_expand(self.count_u8)                 # saturated global counts back into the int32 map
            else:
                if hasattr(e, "count_map_ready"):
                    e.count_map_ready()                       # the evaluation leaves markers; the counts come from the accumulator lines
                dist.all_reduce(self.count)                   # X1 (SUM), exact counts
        e.eval_finish()                                       # E2 (enqueue only)
        # F1: a single GPU never needs P on the host mid-step; with several ranks the host needs the pack length for X2
        self.P, self.pack_len = e.form_active(thres_valid_pixel, sync=multi)
        # X2 in two parts: the per-pixel A22 | b2 rows are final once the active set has been written (F1), so their all-reduce — the
        # bulk of the exchange: 5 doubles per active pixel — runs on the collective's own stream WHILE the Gram kernel (F2) forms
        # A11 | b1; only the 9K^2 + 3K head waits for it.  (async_op: the collective waits for the work enqueued so far on the
        # current stream, and wait() makes the current stream wait for the collective.)
        head = self.pack_len - 5 * self.P if split_x2 else self.pack_len
        part = None
        if split_x2 and self.P:
            part = dist.all_reduce(self.pack[head: self.pack_len], async_op=True)   # X2b (SUM)
        e.form_accumulate(cost_type, a)                       # F2
        if multi:
            dist.all_reduce(self.pack[: head])                # X2a (SUM)
            if part is not None:
                part.wait()
        out = e.form_finish(alpha, download)                  # F3 — the step's host synchronization
        n_inl, self.P = e.last_counts()
        return n_inl, out

    def _exchange_records(self):
        """Records to their pixels' owners (count, pack, all-to-all) -> (recv tensor or None, n_recv).  Round 6: a re-solve of the SAME
        equations with another lambda (after a rejected trial, solver.cpp:340-352: 5 of 7 solves in an LM loop) finds the received records still on the
        owners, in pixel order — one one-word all-reduce (every rank must take the same branch) instead of the exchange: (None, n_recv)."""
        import torch
        e, dist, w, r = self.engine, self.dist, self.world, self.rank
        dev = self.pack.device
        cached = e.solve_shard_cached(r, w) if hasattr(e, "solve_shard_cached") else None
        flag = torch.tensor([1.0 if cached is not None else 0.0], dtype=torch.float64, device=dev)
        if w > 1:
            dist.all_reduce(flag)
        self.last_solve_exchanged = not (float(flag.item()) == float(w))
        if not self.last_solve_exchanged:
            return None, int(cached)
        counts = e.solve_shard_count(w)                                            # records this rank sends to every owner
        table = torch.zeros(w * w, dtype=torch.int64, device=dev)
        table[r * w:(r + 1) * w] = torch.from_numpy(counts).to(dev)
        if w > 1:
            dist.all_reduce(table)                                                 # everybody's counts (disjoint rows)
        table = table.cpu().numpy().reshape(w, w)
        n_send, recv_counts = int(counts.sum()), table[:, r]
        n_recv = int(recv_counts.sum())
        send = torch.empty(max(n_send, 1) * 16, dtype=torch.float64, device=dev)
        recv = torch.empty(max(n_recv, 1) * 16, dtype=torch.float64, device=dev)
        e.solve_shard_pack(w, send)
        if w > 1:
            dist.all_to_all_single(recv[: n_recv * 16], send[: n_send * 16], [int(v) * 16 for v in recv_counts], [int(v) * 16 for v in counts])
        else:
            e.sync(); recv[: n_send * 16] = send[: n_send * 16]; _device_sync(dev)
        return recv, n_recv

    def solveNormalEq(self, lam, fix_first_pose=False, resident_x2=False):
        """LEGM::solveNormalEq (model.cpp:721-792) over all ranks, after iteration(): the sparse A12 factors are time-sharded, a pixel's
        columns are sums over several ranks' records, so the records are first sent to the rank that owns their pixel (contiguous
        ranges of the active set), each rank forms the Schur sums of ITS pixels, one all-reduce of the (3K+1)^2 block follows, the
        Cholesky is replicated and the per-pixel x2 are exchanged.  Three collectives: all-to-all (records), all-reduce (S), all-reduce
        (x2, disjoint supports).  Returns (x1 [3K], x2 [2P]) — identical on every rank.  resident_x2: x2 is returned as the DEVICE tensor
        the all-reduce left on this rank (for ShardedModel.updateMap: no trip through the host)."""
        import torch
        e, dist, w, r = self.engine, self.dist, self.world, self.rank
        dev = self.pack.device
        recv, n_recv = self._exchange_records()
        S = torch.zeros(e.solve_shard_size(), dtype=torch.float64, device=dev)
        _fill_done(dev)
        e.solve_shard_partial(r, w, recv, n_recv, lam, S)
        if w > 1:
            dist.all_reduce(S)
        # A numeric failure is rank-LOCAL (a 2x2 block of A22m that is not positive definite shows up on its pixel's owner only): the
        # decision to raise must be global, or one rank leaves the protocol while the others wait in the next collective.  The status rides
        # as one more element of the x2 all-reduce, which every rank always executes; then all ranks raise together.
        # x1 rides along too (rank 0's copy, zeros from the others): the replicated factorisation combines partial tiles with LDS atomics,
        # so its result is equal across ranks only to rounding — every rank must apply the SAME pose update or the replicas drift apart.
        n2, n1 = 2 * max(self.P, 1), 3 * int(e.K)
        x2 = torch.zeros(n2 + 1 + n1, dtype=torch.float64, device=dev)      # [x2 (2P) | status | x1 (3K)]
        _fill_done(dev)
        x1, failure = None, None
        try:
            x1 = e.solve_shard_finish(r, w, recv, n_recv, lam, fix_first_pose, S, x2)
        except Exception as exc:   # noqa: BLE001
            if getattr(exc, "status", None) != _ERR_NUMERIC:
                raise
            failure = exc
            x2[n2] = 1.0
        if r == 0 and x1 is not None:
            x2[n2 + 1:] = torch.from_numpy(np.ascontiguousarray(x1)).to(dev)
        if w > 1:
            dist.all_reduce(x2)
        _device_sync(dev)
        if float(x2[n2].item()) != 0.0:
            raise failure if failure is not None else EmbaError(_ERR_NUMERIC, "the damped normal equations are not positive definite on another rank")
        x1 = x2[n2 + 1:].cpu().numpy().copy()
        return x1, (x2[: 2 * self.P] if resident_x2 else x2[: 2 * self.P].cpu().numpy())


    # (method of ShardedLEGM, defined below the class body's other solvers for readability)


def _sharded_solve_cg(self, lam, fix_first_pose=False, max_iter=100, tol=1e-6, resident_x2=False):
    """LEGM::solveNormalEqCG (model.cpp:794-840; solver.cpp:190-202 selects it with use_CG) over all ranks — round 6.  The pixels are sharded by owner as in
    solveNormalEq (same record exchange, skipped on a re-solve); Eigen's loop (ConjugateGradient.h:28-88: diagonal preconditioner, zero initial guess, 100
    iterations, tolerance 1e-6 on |r| / |b|) runs on every rank on scalars read from the SAME all-reduced sums: one all-reduce of 3K + 2 doubles per
    application of the matrix, one of 2 per iteration.  Returns (x1, x2, iterations, error) — identical on every rank."""
    import torch
    e, dist, w, r = self.engine, self.dist, self.world, self.rank
    dev = self.pack.device
    recv, n_recv = self._exchange_records()
    n = 3 * int(e.K)
    red = torch.zeros(e.cg_shard_size(), dtype=torch.float64, device=dev)
    _fill_done(dev)
    tiny = float(np.finfo(np.float64).tiny)

    def reduce(t):
        if w > 1:
            dist.all_reduce(t)
        _device_sync(dev)

    e.cg_shard_begin(r, w, recv, n_recv, lam, fix_first_pose, red)
    reduce(red)
    rhs2, abs_new = (float(v) for v in red[n:n + 2].cpu())
    it, err = 0, 0.0
    if rhs2 != 0.0:
        thr = max(tol * tol * rhs2, tiny)
        rn2 = rhs2
        if rn2 >= thr:
            while it < max_iter:
                e.cg_shard_apply(red)
                reduce(red)
                alpha = abs_new / e.cg_shard_pt(red)
                e.cg_shard_update(alpha, red)
                tail = red[n:n + 2]
                reduce(tail)
                rn2, abs_next = (float(v) for v in tail.cpu())
                if rn2 < thr:
                    break
                beta = abs_next / abs_new
                abs_new = abs_next
                e.cg_shard_direction(beta)
                it += 1
        err = float(np.sqrt(rn2 / rhs2))
    x2 = torch.zeros(2 * max(self.P, 1), dtype=torch.float64, device=dev)
    _fill_done(dev)
    x1 = e.cg_shard_end(x2)
    reduce(x2)
    return np.asarray(x1).copy(), (x2[: 2 * self.P] if resident_x2 else x2[: 2 * self.P].cpu().numpy()), it, err


ShardedLEGM.solveNormalEqCG = _sharded_solve_cg


class ShardedModel:
    """The model interface emba_amd.solver.solve_time_window drives (EMBA::solveTimeWindow, solver.cpp:63-353), over all ranks: every rank
    runs the same loop on its time shard and takes the same decisions, because everything a decision depends on is reduced — the data cost
    is summed over the ranks, the normal equations and the Schur solve are the sharded ones, the map is replicated and updated with the
    same x2 everywhere.  Use with resident=True.  `legm` is this rank's emba_amd.LEGM (already wrapped by `sharded`'s engine)."""

    def __init__(self, sharded, legm):
        self.sh, self.m = sharded, legm
        self.cost = ("quadratic", 0.0)
        self._thres = None
        self.H, self.W = legm.H, legm.W

    def set_events(self, events):
        self.sh.set_events(events)

    def set_cost(self, cost_type="quadratic", a=0.0):
        self.cost = (cost_type, a)

    def upload_map(self, Gx, Gy):
        self.m.upload_map(Gx, Gy)

    def eval_launch(self, traj):
        self.sh.evaluate(traj, *self.cost)

    def eval_finish(self):
        self.sh.engine.eval_finish()

    def dataCost(self, cost_type="quadratic", a=0.0):
        import torch
        local = self.m.dataCost(cost_type, a)                 # this rank's measurements only (halo entries are not counted)
        if self.sh.world == 1:
            return local
        t = torch.tensor([local], dtype=torch.float64, device=self.sh.pack.device)
        self.sh.dist.all_reduce(t)
        return float(t.item())

    def regCost(self, alpha):
        return self.m.regCost(alpha)                          # replicated map: identical on every rank

    def form_active(self, thres):
        self._thres = thres

    def form_accumulate(self, cost_type="quadratic", a=0.0):
        self.cost = (cost_type, a)

    def form_finish(self, alpha):
        self.sh.form(self._thres, alpha, *self.cost)

    keeps_equations_on_reject = True
    supports_resident_x2 = True     # solver.solve_time_window: x2 stays a device tensor between solveNormalEq and updateMap

    def solveNormalEq(self, lam, fix_first_pose=False, resident_x2=False):
        return self.sh.solveNormalEq(lam, fix_first_pose, resident_x2=resident_x2)

    def solveNormalEqCG(self, lam, fix_first_pose=False, resident_x2=False):
        x1, x2, _, _ = self.sh.solveNormalEqCG(lam, fix_first_pose, resident_x2=resident_x2)
        return x1, x2

    def updateMap(self, x2, damping):
        if hasattr(x2, "data_ptr"):            # the tensor of solveNormalEq(resident_x2=True)
            if x2.is_cuda and getattr(self.m, "supports_resident_x2", False):
                self._x2_keep = x2             # (kept alive until the next update: the copy is enqueued, not finished)
                self.m.updateMap(int(x2.data_ptr()) if x2.numel() else None, damping)      # (no active pixel: nothing to apply)
            else:
                self.m.updateMap(x2.cpu().numpy(), damping)
        else:
            self.m.updateMap(x2, damping)

    def acceptMap(self):
        self.m.acceptMap()

    def rejectMap(self):
        self.m.rejectMap()

    def downloadMap(self):
        return self.m.downloadMap()


class HipEngine:
    """Adapter: emba_amd.LEGM phase calls + torch CUDA tensors as the exchange buffers (product path)."""

    step_fast = True     # the ranks' forms run as resident steps where they can (emba_step_form_active); False: the sweeping forms (A/B)
    x2_split = None      # None: auto (from 3 M events per rank); 0 / 1: exchange 2 in one piece / split (x2_split_pays)

    def __init__(self, legm, check_stream=True):
        """check_stream: the collectives of torch.distributed run on torch's CURRENT stream, the kernels on the LEGM's stream; unless
        the two are the same stream nothing orders a kernel against the all-reduce that follows it.  Pass False only with a `dist`
        that synchronises by itself (the thread stand-in of tests/test_gpu_sharded.py drains the engine's stream first)."""
        self.m = legm
        self.check_stream = check_stream

    @property
    def K(self):
        return self.m.K

    def bind_exchange(self, count_tensor, pack_tensor):
        assert count_tensor.is_cuda and pack_tensor.is_cuda and count_tensor.is_contiguous() and pack_tensor.is_contiguous()
        if self.check_stream:
            import torch
            cur = torch.cuda.current_stream(count_tensor.device).cuda_stream
            if getattr(self.m, "stream", None) != cur:
                raise RuntimeError("HipEngine: create the LEGM with stream=torch.cuda.current_stream().cuda_stream (a non-default "
                                   "torch stream): kernels and RCCL collectives must share one stream")
        self.m.bind_exchange_buffers(count_tensor.data_ptr(), pack_tensor.data_ptr(), pack_tensor.numel())

    def set_events(self, events, halo):
        self.m.set_events(events, halo)

    def upload_map(self, Gx, Gy):
        self.m.upload_map(Gx, Gy)

    def declare_cost(self, cost_type, a):
        """emba_set_cost: evaluations accumulate the per-pixel sums with this cost's weights, so the A22 | b2 rows written by
        form_active are final (form_accumulate only adds A11 | b1) and their all-reduce may start before the Gram kernel."""
        self.m.set_cost(cost_type, a)
        return True

    def x2_split_pays(self, n_max):
        """Splitting exchange 2 hides the bulk of it behind the Gram kernel but adds one small collective: worth it from a few million
        events per rank (Gram kernel >= ~100 us against a collective's tens of microseconds of latency); HipEngine.x2_split = 0 / 1 overrides
        (the same on every rank: it changes the sequence of collectives).
        n_max: events of the LARGEST shard — the same number on every rank (ShardedLEGM.set_events)."""
        if self.x2_split is not None:
            return bool(self.x2_split)
        return n_max >= 3_000_000

    def eval_launch(self, traj):
        self.m.eval_launch(traj)

    def eval_finish(self):
        self.m.eval_finish(sync=False)

    def form_active(self, thres, sync=True):
        return self.m.form_active(thres, sync=sync)

    def last_counts(self):
        return self.m.last_counts()

    def step(self, traj, thres, alpha, cost_type, a):
        return self.m.step(traj, thres, alpha, cost_type, a)

    def count_map_ready(self):
        self.m.count_map_ready()

    def count_compress(self, u8_tensor, cap):
        self.m.count_compress(u8_tensor.data_ptr(), cap)

    def count_expand(self, u8_tensor):
        self.m.count_expand(u8_tensor.data_ptr())

    def step_form_active(self, thres, u8_tensor):
        self.m.step_form_active(thres, u8_tensor.data_ptr())

    def form_accumulate(self, cost_type, a):
        self.m.form_accumulate(cost_type, a)

    def form_finish(self, alpha, download):
        return self.m.form_finish(alpha, download)

    def sync(self):
        self.m.sync()

    def solve_shard_size(self):
        return self.m.solve_shard_size()

    def solve_shard_count(self, n_ranks):
        return self.m.solve_shard_count(n_ranks)

    def solve_shard_pack(self, n_ranks, send):
        self.m.solve_shard_pack(n_ranks, send.data_ptr())

    def solve_shard_cached(self, rank, n_ranks):
        return self.m.solve_shard_cached(rank, n_ranks)

    # sharded solveNormalEqCG: the per-rank steps around ShardedLEGM.solveNormalEqCG's collectives
    def cg_shard_size(self):
        return self.m.cg_shard_size()

    def cg_shard_begin(self, rank, n_ranks, recv, n_recv, lam, fix_first_pose, red):
        self.m.cg_shard_begin(rank, n_ranks, recv.data_ptr() if recv is not None else None, n_recv, lam, fix_first_pose, red.data_ptr())

    def cg_shard_apply(self, red):
        self.m.cg_shard_apply(red.data_ptr())

    def cg_shard_pt(self, red):
        return self.m.cg_shard_pt(red.data_ptr())

    def cg_shard_update(self, alpha, red):
        self.m.cg_shard_update(alpha, red.data_ptr())

    def cg_shard_direction(self, beta):
        self.m.cg_shard_direction(beta)

    def cg_shard_end(self, x2):
        return self.m.cg_shard_end(x2.data_ptr())

    def solve_shard_partial(self, rank, n_ranks, recv, n_recv, lam, S):      # recv None: the rank's cached records (solve_shard_cached)
        self.m.solve_shard_partial(rank, n_ranks, recv.data_ptr() if recv is not None else None, n_recv, lam, S.data_ptr())

    def solve_shard_finish(self, rank, n_ranks, recv, n_recv, lam, fix_first_pose, S, x2):
        return self.m.solve_shard_finish(rank, n_ranks, recv.data_ptr() if recv is not None else None, n_recv, lam, fix_first_pose, S.data_ptr(), x2.data_ptr())
