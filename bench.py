#!/usr/bin/env python3
"""bench.py — events/s through warp + Jacobian + J^T J build (BASELINE.json metric) on N MI355X GPUs of one node.

A "step" is one pass of the hot path over the resident workload: evaluateDataError(eval_deriv=true) + formNormalEq +
applyL2Reg (reference solver.cpp:75/251 + :114-130) with events, map planes and LUT already in HBM when the timed region
starts.  N=1: the BASELINE configuration (synthetic shapes-like: 1 M events, 240x180 sensor, 1024x2048 panorama, K=21,
seed 20240907).  N>1 (launched by torch.distributed.run, one rank per GPU): weak scaling — N x 1 M events over the same
1 s window, sharded by time range with a per-pixel halo; per step one all-reduce of the count map (as saturated bytes) and one
of the fp64 normal-equation pack over RCCL (emba_amd/sharded.py).

Prints ONE JSON line on rank 0.  The `roofline` object prices the dominant kernel (emba_warp_residual_kernel) with
SURVEY §8d's 244 algorithmic bytes per event against the 8 TB/s HBM3E peak, its duration measured live with HIP events on
the stream the kernel runs on.  `cpu_baseline` times the CPU oracle (single-threaded port of the reference algorithm) on the
GPU box's host, rank 0, N=1 only; it is reported, not measured-as-product.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_EVENT = 244
HBM_PEAK_GBS = 8000.0               # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
EVENTS_PER_GPU = 1_000_000


def pmc_traffic(workload_desc):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (scripts/profile.sh +
    scripts/summarize_profiles.py: separate FETCH_SIZE / WRITE_SIZE passes, gfx950 correction 2*FETCH+WRITE).  Counters
    cannot be collected from inside this process, so the latest committed summary for the same workload is reported."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("workload") == workload_desc:
            best = d
    return best


def host_cores():
    """Cores this process may use on the GPU box: CPU affinity, capped by the cgroup CPU quota when one is set."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(int(q) / int(per))))
    except Exception:
        pass
    return n


def _time_oracle(O, w, budget_s, max_passes):
    o = O.OracleLEGM(w.sensor_w, w.sensor_h, w.pano_w, w.pano_h, w.lut, w.C_th)
    ev = w.events
    times = []
    t_all = time.perf_counter()
    while True:
        t0 = time.perf_counter()
        ep, nem = o.evaluate_data_error(w.traj.knots_xyzw, w.traj.t0_ns, w.traj.dt_ns, w.Gx, w.Gy, ev.x, ev.y, ev.polarity, ev.t_ns)
        ne = o.form_normal_eq(ep, w.K, nem, w.thres_valid_pixel)
        o.apply_l2(ne, w.alpha, w.Gx, w.Gy)
        times.append(time.perf_counter() - t0)
        if time.perf_counter() - t_all > budget_s or len(times) >= max_passes:
            break
    return float(np.median(times[1:] if len(times) > 1 else times)), len(times)


def cpu_baseline(w, budget_s=10.0):
    """The CPU oracle on the same workload on the GPU box's host cores, both ways SURVEY §8d asks for: "ref" = ONE thread in the
    reference's evaluation order (the reference is single-threaded) — the headline `cpu_baseline` — and "omp" = the same arithmetic on
    all the cores this process may use (core count stated).  Bounded to ~budget_s of wall time each."""
    from oracle import oracle as O
    ev = w.events
    O.set_threads(1)
    med1, n1 = _time_oracle(O, w, budget_s, 25)
    cores = max(1, min(host_cores(), O.max_threads()))
    O.set_threads(cores)
    try:
        medn, nn = _time_oracle(O, w, budget_s, 40)
    finally:
        O.set_threads(1)
    native = cpu_baseline_native(budget_s / 2)
    return {"value": ev.size() / med1, "unit": "events/s", "cores": 1, "kind": "port", "march_native": native,
            "sample": f"full workload ({ev.size()} events), {n1} passes, median pass {med1 * 1e3:.1f} ms, "
                      f"host nproc={os.cpu_count()}, usable cores={host_cores()}",
            "all_cores": {"value": ev.size() / medn, "unit": "events/s", "cores": cores, "kind": "port (OpenMP mode of the oracle)",
                          "sample": f"full workload ({ev.size()} events), {nn} passes, median pass {medn * 1e3:.1f} ms"}}


def cpu_baseline_native(budget_s):
    """SURVEY §8d's protocol asks for `-O3 -march=native`; the checker library that travels with the repository is built without it (one build for the
    authoring container and for the GPU box's host, a different CPU).  Here the same source is compiled ONCE MORE on this host with -march=native (into
    a temporary directory; FMA contraction stays off — the restatement's bit-exactness needs that) and timed in a child process on one thread.
    Reported beside the headline figure, never instead of it; None if there is no compiler."""
    import shutil
    import subprocess
    import tempfile
    if not shutil.which("gcc"):
        return None
    d = tempfile.mkdtemp(prefix="emba_oracle_native_")
    lib = os.path.join(d, "libemba_oracle_native.so")
    try:
        subprocess.check_call(["gcc", "-O3", "-march=native", "-fPIC", "-std=c11", "-ffp-contract=off", "-fopenmp", "-shared",
                               os.path.join(ROOT, "oracle", "emba_oracle.c"), "-o", lib, "-lm"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        code = ("import sys, json; sys.path.insert(0, %r); import bench; from oracle import oracle as O; from emba_amd.synth import make_workload; "
                "w = make_workload(); O.set_threads(1); m, n = bench._time_oracle(O, w, %f, 12); print(json.dumps({'median_s': m, 'passes': n, 'events': w.events.size()}))" % (ROOT, budget_s))
        env = dict(os.environ); env["EMBA_ORACLE_LIB"] = lib; env["OMP_NUM_THREADS"] = "1"
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, env=env)
        j = json.loads(r.stdout.strip().splitlines()[-1])
        return {"value": j["events"] / j["median_s"], "unit": "events/s", "cores": 1, "kind": "port, rebuilt on this host with -O3 -march=native -ffp-contract=off",
                "sample": f"full workload, {j['passes']} passes, median pass {j['median_s'] * 1e3:.1f} ms"}
    except Exception as e:   # noqa: BLE001
        return {"error": repr(e)[:200]}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def launch_ranks(n):
    """One rank per GPU as a child `python -m torch.distributed.run` (rendezvous on 127.0.0.1, a free port), same arguments."""
    import socket
    import subprocess
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--events-per-gpu", type=int, default=EVENTS_PER_GPU)
    ap.add_argument("--pano-h", type=int, default=1024)
    ap.add_argument("--knots", type=int, default=21)
    ap.add_argument("--sensor", default="240x180", help="sensor WxH (focal scaled to keep the field of view)")
    ap.add_argument("--data", choices=["uniform", "scene"], default="uniform",
                    help="uniform: SURVEY §8d's i.i.d. events (the BASELINE workload).  scene: events an ideal event camera fires while "
                         "rotating in front of an analytic scene (edge-clustered, polarity-consistent; emba_amd.synth.simulate_events)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--with-ep", action="store_true",
                    help="(always on since round 4) the step WITH the compaction of the residuals into the reference-order ep vector (what the one-shot "
                         "drop-in entry point emba_eval_data_error always produces) is timed in an extra block; reported as config.with_ep_ms_per_step, never as value")
    ap.add_argument("--shard-of", type=int, default=1,
                    help="time ONE rank's shard of a window sharded over this many GPUs, on one GPU and without collectives: the global stream has "
                         "shard-of x events-per-gpu events, the rank holds its time range + per-pixel halo (what each GPU of configs 4 / 5 computes)")
    ap.add_argument("--shard-rank", type=int, default=0)
    ap.add_argument("--yaw-rate", type=float, default=0.5)
    ap.add_argument("--one-device", action="store_true",
                    help="rehearsal of the N-rank launch on a ONE-GPU box: every rank uses device 0 and the collectives go through gloo "
                         "(RCCL refuses two ranks on one device); exercises the launcher, the sharding and the protocol, not xGMI")
    ap.add_argument("--force-collectives", action="store_true",
                    help="rehearsal on ONE GPU: initialise RCCL with world_size 1 and run both all-reduces of the sharded protocol")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "RANK" not in os.environ:
        # `python bench.py --gpus N` typed as is: start one rank per GPU ourselves.  This process has not touched the GPU (torch is not
        # even imported yet) and never will: it starts torch.distributed.run as a CHILD process and passes its exit code on.
        sys.exit(launch_ranks(args.gpus))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} != WORLD_SIZE {world}")

    import torch
    import torch.distributed as dist

    assert torch.cuda.is_available(), "bench.py needs the MI355X (no CPU fallback)"
    if args.one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_collectives
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        if args.one_device:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)

    from emba_amd import LEGM
    from emba_amd.sharded import HipEngine, ShardedLEGM
    from emba_amd.synth import make_workload

    n_total = args.events_per_gpu * world * max(args.shard_of, 1)
    sw, sh_ = (int(v) for v in args.sensor.lower().split("x"))
    if args.data == "scene":
        from emba_amd.synth import make_scene_stream
        w = make_scene_stream(n_total, pano_h=args.pano_h, K=args.knots, sensor=(sw, sh_), focal=200.0 * sw / 240.0)
        n_total = w.events.size()
    else:
        w = make_workload(n_events=n_total, pano_h=args.pano_h, K=args.knots, sensor=(sw, sh_), focal=200.0 * sw / 240.0, yaw_rate=args.yaw_rate)
    npix = w.pano_h * w.pano_w

    # One explicit (non-null) HIP stream for our kernels AND for torch/RCCL, so that launches and collectives are ordered.
    tstream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(tstream)
    stream = tstream.cuda_stream
    m = LEGM(w.sensor_w, w.sensor_h, w.lut, w.C_th, w.pano_w, w.pano_h, device=local_rank, stream=stream)
    count_t = torch.zeros(npix, dtype=torch.int32, device=dev)
    pack_t = torch.zeros(9 * w.K * w.K + 3 * w.K + 5 * npix, dtype=torch.float64, device=dev)

    class _Dist:  # torch.distributed or a 1-rank stand-in with the same three calls
        @staticmethod
        def get_rank(): return rank
        @staticmethod
        def get_world_size(): return world
        @staticmethod
        def all_reduce(t, async_op=False): return dist.all_reduce(t, async_op=async_op)
        @staticmethod
        def all_to_all_single(out, inp, out_splits, in_splits): dist.all_to_all_single(out, inp, out_splits, in_splits)

    count_u8 = torch.zeros(npix, dtype=torch.uint8, device=dev)   # exchange 1 travels as saturated bytes (emba_amd/sharded.py)
    sh = ShardedLEGM(HipEngine(m), _Dist, count_t, pack_t, w.sensor_w, count_u8)
    sh.force_collectives = args.force_collectives
    t_set = time.perf_counter()
    if args.shard_of > 1:
        from emba_amd.sharded import shard_events
        assert world == 1, "--shard-of times one rank's shard on one GPU"
        local, halo = shard_events(w.events, w.sensor_w, args.shard_rank, args.shard_of)
        sh.engine.set_events(local, halo); sh.n_local = local.size(); sh.n_max = local.size()
        n_total = local.size()
    else:
        local = sh.set_events(w.events)
    m.upload_map(w.Gx, w.Gy)                                 # HBM-resident before the timed region
    m.sync()
    t_set = time.perf_counter() - t_set

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def step():
        return sh.iteration(w.traj, w.thres_valid_pixel, w.alpha)

    for _ in range(args.warmup):
        step()
    # beyond the caller's W steps: keep stepping (untimed) until >= 50 ms of GPU work has run, so that a short timed region
    # (20 steps = 2.6 ms at 1 M events) does not sit on the clock ramp of a cold chip.  The NUMBER of extra steps is one decision
    # for all ranks (every step issues collectives: a rank-local wall-clock loop would leave the ranks a step apart): three steps
    # are timed between barriers, the slowest rank's time is all-reduced, and everybody runs the same count.
    barrier()
    t_w = time.perf_counter()
    for _ in range(3):
        step()
    barrier()
    t_w = (time.perf_counter() - t_w) / 3
    if use_dist:
        tt = torch.tensor([t_w], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        t_w = float(tt.item())
    n_extra = int(min(max(np.ceil(0.05 / max(t_w, 1e-6)), 1), 2000))
    for _ in range(n_extra):
        step()
    barrier()
    # ... and until the step time has SETTLED, and at least 1000 steps have run in all.  Round 4 (scripts/r04_dbg.py): the HIP runtime stalls the
    # launching thread ONCE per process for 32-68 ms at a fixed number of kernel launches — step 848-850 of this loop, whatever the step's size or
    # the stream (and for ~1.2 ms at step 460-462); nothing like it in the following 3000 steps.  A 20-step timed region that happens to contain
    # it reads 3 ms per step instead of 0.1.  Blocks of steps are timed between barriers until two consecutive blocks agree within 3 % AND the
    # thousand is complete; the block time is all-reduced (MAX), so every rank runs the same count.
    n_blk = int(min(max(np.ceil(0.02 / max(t_w, 1e-6)), 8), 400))
    prev, done = None, args.warmup + 3 + n_extra
    for _ in range(200):
        t_b = time.perf_counter()
        for _ in range(n_blk):
            step()
        barrier()
        done += n_blk
        t_b = (time.perf_counter() - t_b) / n_blk
        if use_dist:
            tt = torch.tensor([t_b], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            t_b = float(tt.item())
        if prev is not None and abs(t_b - prev) <= 0.03 * prev and done >= 1000:
            break
        prev = t_b
    # Kernel durations by HIP events on the kernels' stream, sampled on every 8th step of the timed region (at most 16 samples), each sample in
    # its own set of events that is read AFTER the loop: an event record opens a bubble of a few us in front of the next kernel, and reading one
    # back inside the loop would make the host wait for the Gram kernel — timing every step would distort the very throughput being measured.
    every = max(8, (args.steps + 15) // 16)
    slots = []
    t0 = time.perf_counter()
    for i in range(args.steps):
        timed = (i % every == 0)
        if timed:
            slots.append(len(slots))
        m.enable_kernel_timing(timed, slots[-1] if timed else 0)
        if os.environ.get("BENCH_DEBUG"):
            t_s = time.perf_counter(); n_inl, _ = step(); t_s = time.perf_counter() - t_s
            if t_s > 1e-3: print(f"[bench debug] step {i} took {t_s * 1e3:.2f} ms (timed={timed})", file=sys.stderr)
        else:
            n_inl, _ = step()
    barrier()
    elapsed = time.perf_counter() - t0
    m.enable_kernel_timing(False)
    warp_ms, accum_ms = [], []
    for sl in slots:
        a, b = m.kernel_ms_slot(sl)
        warp_ms.append(a); accum_ms.append(b)
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # the step WITH the residuals compacted into the reference-order ep vector, as one more untimed block of steps (like the exchanges below):
    # reported as config.with_ep_ms_per_step in every line, never as `value`
    with_ep_ms = None
    if True:
        barrier()
        t_e = time.perf_counter()
        for _ in range(args.steps):
            step()
            m.compact_ep()
        barrier()
        with_ep_ms = (time.perf_counter() - t_e) / args.steps * 1e3

    # the exchanges alone (same buffers, same sizes, same stream), untimed w.r.t. `value`: what a step spends in collectives
    coll_ms = None
    if use_dist:
        plen = int(sh.pack_len) if sh.pack_len else pack_t.numel()
        barrier()
        t_c = time.perf_counter()
        for _ in range(10):
            dist.all_reduce(count_u8)
            dist.all_reduce(pack_t[:plen])
        barrier()
        coll_ms = (time.perf_counter() - t_c) / 10 * 1e3

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = n_total / (elapsed / args.steps)
        wk = float(np.mean(warp_ms))
        n_launch = local.size()
        # (a shard-of-N run times one rank's part of the stream: its traffic summary is keyed by the shard, not by the whole window's description)
        wkey = w.describe() if args.shard_of == 1 else f"{w.describe()} shard {args.shard_rank} of {args.shard_of}"
        tr = pmc_traffic(wkey) if world == 1 else None
        achieved = ALGO_BYTES_PER_EVENT * n_launch / (wk * 1e-3) / 1e9
        counter = (tr["hbm_bytes_per_launch"] / (wk * 1e-3) / 1e9) if tr else None   # the bytes the counters saw, over the same time
        out = {
            "metric": "events/sec through warp+Jacobian+JtJ build, 1M events, 1024x2048 pano",
            "value": value, "unit": "events/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic" if args.data == "uniform" else "synthetic (simulated scene)",
            "config": {"workload": w.describe(), "events_per_gpu": args.events_per_gpu, "total_events": n_total,
                       "thres_valid_pixel": w.thres_valid_pixel, "alpha": w.alpha, "cost": "quadratic",
                       "step": "evaluateDataError(eval_deriv)+formNormalEq+applyL2Reg, inputs resident in HBM; residuals stay per event in HBM "
                               "(the host API's compacted ep vector is produced when it is asked for)",
                       "parallelism": f"time-sharded x{world}" if world > 1 else (f"shard {args.shard_rank} of {args.shard_of} of the window, one GPU, no collectives" if args.shard_of > 1 else "single GPU"),
                       "events_per_rank": int(local.size()), "collectives_ms_per_step": coll_ms, "with_ep_ms_per_step": with_ep_ms,
                       "backend": ("gloo, all ranks on device 0 (rehearsal)" if args.one_device else "rccl") if use_dist else None,
                       "inliers_rank0": int(n_inl), "active_pixels": int(sh.P), "set_events_s": round(t_set, 3),
                       "setup": m.setup_info()},
            "roofline": {"bound": "hbm", "kernel": "emba_warp_tiled_kernel" if m.setup_info()["tile_order"] else "emba_warp_residual_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": (tr["hbm_bytes_per_launch"] if tr else None), "traffic_unit": "bytes/launch (rocprofv3 PMC, profiles/)",
                         "counter_GBs": counter, "counter_frac": (counter / HBM_PEAK_GBS if counter else None),
                         "algorithmic_bytes_per_launch": ALGO_BYTES_PER_EVENT * n_launch,
                         "bytes_per_event": ALGO_BYTES_PER_EVENT, "events_per_launch": n_launch, "kernel_ms": wk,
                         "accumulate_kernel_ms": float(np.mean(accum_ms)),
                         "path_frac": ALGO_BYTES_PER_EVENT * n_launch / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(w)
        print(json.dumps(out))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
