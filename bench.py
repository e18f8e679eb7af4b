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


def _read(path):
    try:
        return open(path).read().strip()
    except Exception:
        return None


def device_power_state(pci_id=None):
    """What the card itself reports through sysfs while the bench runs (VERDICT r4 #1c): current sclk / mclk / fclk level of pp_dpm_*, power cap
    and power draw.  pci_id (hipDeviceGetPCIBusId of the context's device) names the card; without it every card of the host is listed."""
    import glob
    import re
    dirs = [f"/sys/bus/pci/devices/{pci_id.lower()}"] if pci_id else sorted(glob.glob("/sys/class/drm/card[0-9]*/device"))[:8]
    out = {}
    for d in dirs:
        ent = {}
        for key, fn in (("sclk_mhz", "pp_dpm_sclk"), ("mclk_mhz", "pp_dpm_mclk"), ("fclk_mhz", "pp_dpm_fclk")):
            t = _read(os.path.join(d, fn))
            if t:
                cur = [ln for ln in t.splitlines() if ln.rstrip().endswith("*")] or t.splitlines()[-1:]
                mm = re.search(r"(\d+)\s*Mhz", cur[0], re.I)
                ent[key] = int(mm.group(1)) if mm else cur[0]
        for hw in glob.glob(os.path.join(d, "hwmon", "hwmon*")):
            for key, fn in (("power_cap_w", "power1_cap"), ("power_avg_w", "power1_average"), ("power_w", "power1_input")):
                t = _read(os.path.join(hw, fn))
                if t and t.lstrip("-").isdigit():
                    ent[key] = round(int(t) * 1e-6, 1)
        if ent:
            out[os.path.basename(d) if pci_id else os.path.basename(os.path.dirname(d))] = ent
    if pci_id and out:
        return next(iter(out.values()))
    if not out:
        import shutil
        import subprocess
        smi = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
        try:
            r = subprocess.run([smi, "--showclocks", "--showpower", "--showmaxpower", "--json"], capture_output=True, text=True, timeout=30)
            out = {"rocm_smi": json.loads(r.stdout)} if r.returncode == 0 and r.stdout.strip().startswith("{") else {"rocm_smi_error": (r.stderr or r.stdout)[-200:]}
        except Exception as e:   # noqa: BLE001
            out = {"error": repr(e)[:200]}
    return out


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
            "all_cores_value": ev.size() / medn, "all_cores": cores, "march_native_value": (native or {}).get("value"),
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
    ap.add_argument("--no-with-ep", action="store_true",
                    help="skip the extra (untimed w.r.t. value) blocks that time the step with / without the compaction of the residuals into the "
                         "reference-order ep vector (config.with_ep_ms_per_step, config.no_ep_ms_per_step)")
    ap.add_argument("--long-steps", type=int, default=300,
                    help="steps of the second, longer timed block behind the K contractual ones (ms_per_step_long; every 16th of its steps is sampled with HIP "
                         "events around ALL of its launches); 0 skips it")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE",
                    help="emba_set_option on the context (A/B comparisons: order=1, step_gather=0, step_ep=0 ...); results do not depend on any of them")
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

    # ONE JSON line on stdout, whatever the libraries underneath print: RCCL writes a five-line version banner to STDOUT when its first communicator comes up
    # (seen with --force-collectives, round 5).  From here on file descriptor 1 is the process's stderr; the JSON line goes to the saved descriptor.
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

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
    for kv in args.opt:
        name, _, val = kv.partition("=")
        m.set_option(name.strip(), int(val))
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
    # (the thousand-step floor is for SHORT steps — the stall sits at a launch count, and a 20-step region of 0.1-ms steps is what it ruins; at
    # >= 1 ms per step the floor is what 1 s of stepping gives (a 100 M-event step is 7 ms: the stall is < 1 % of a 4-step region's time per step there,
    # and under rocprofv3 --pmc a thousand serialised steps would not fit the profile scripts' timeouts).  All of it is reported: config.settle.)
    n_blk = int(min(max(np.ceil(0.02 / max(t_w, 1e-6)), 8), 400))
    floor_steps = 1000 if t_w < 1e-3 else int(min(1000, max(3 * n_blk, np.ceil(1.0 / t_w))))
    prev, done, slowest_blk = None, args.warmup + 3 + n_extra, 0.0
    t_settle = time.perf_counter()
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
        slowest_blk = max(slowest_blk, t_b)
        if prev is not None and abs(t_b - prev) <= 0.03 * prev and done >= floor_steps:
            break
        prev = t_b
    settle = {"untimed_steps": int(done), "floor_steps": int(floor_steps), "block": int(n_blk), "wall_s": round(time.perf_counter() - t_settle, 3),
              "slowest_block_ms_per_step": slowest_blk * 1e3, "last_block_ms_per_step": t_b * 1e3}
    # Kernel durations by HIP events on the kernels' stream, sampled on every 8th step of the timed region (at most 16 samples), each sample in
    # its own set of events that is read AFTER the loop: an event record opens a bubble of a few us in front of the next kernel, and reading one
    # back inside the loop would make the host wait for the Gram kernel — timing every step would distort the very throughput being measured.
    every = max(8, (args.steps + 15) // 16)
    slots = []
    t0 = time.perf_counter()
    for i in range(args.steps):
        timed = (i % every == every // 2)      # (not step 0: the first step of the region starts on an idle device; K = 20: steps 4 and 12)
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

    # Round 5: on one GPU the step PRODUCES the reference-order ep vector (what evaluateDataError returns, model.cpp:256) — compacted by tail blocks
    # of its Gram launch (option step_ep) — so `value` is the step with ep.  Two more untimed blocks say what that costs: the same step with the option
    # off (config.no_ep_ms_per_step: what rounds 1-4 reported as `value`), and, where the step does not produce ep itself (several ranks; windows too
    # long for the tail form), the step followed by the stand-alone compaction (config.with_ep_ms_per_step).
    with_ep_ms, no_ep_ms = None, None
    ep_in_step = (world == 1 and not args.force_collectives and m.get_option("step_ep") == 1 and m.get_option("ep_valid") == 1)
    if not args.no_with_ep:
        barrier()
        t_e = time.perf_counter()
        for _ in range(args.steps):
            step()
            m.compact_ep()          # (a no-op where the step has produced ep already)
        barrier()
        with_ep_ms = (time.perf_counter() - t_e) / args.steps * 1e3
        if ep_in_step:
            m.set_option("step_ep", 0)
            for _ in range(3):
                step()
            barrier()
            t_e = time.perf_counter()
            for _ in range(args.steps):
                step()
            barrier()
            no_ep_ms = (time.perf_counter() - t_e) / args.steps * 1e3
            m.set_option("step_ep", 1)
            step()

    # a second, LONGER block of the same step (ms_per_step_long), in which every 16th step is sampled with HIP events around ALL of
    # its launches — consecutive events, so the four intervals tile the sampled step's device time — plus what an event bracket reads around
    # an EMPTY kernel on this box (bracket_overhead_us), the shader clock a probe kernel measures right behind the block, and the card's own
    # report of its clocks and power cap.  None of it touches `value`.
    long_ms, kern_all, bracket_us, clocks, power_state = None, None, None, None, None
    if args.long_steps > 0:
        n_long = int(min(args.long_steps, max(32, np.ceil(3.0 / max(t_b, 1e-6)))))        # at most ~3 s of stepping
        try:
            pci = m.pci_bus_id()
        except Exception:   # noqa: BLE001
            pci = None
        m.kernel_timing_all(True)
        lslots, psamples = [], []
        barrier()
        t_l = time.perf_counter()
        for i in range(n_long):
            timed = (i % 16 == 8) and len(lslots) < 16
            if timed:
                lslots.append(len(lslots))
            m.enable_kernel_timing(timed, lslots[-1] if timed else 0)
            step()
        barrier()
        long_ms = (time.perf_counter() - t_l) / n_long * 1e3
        if world > 1:
            tt = torch.tensor([long_ms], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            long_ms = float(tt.item())
        m.enable_kernel_timing(False)
        # the card's own report WHILE it steps: a short block of its own behind the long one (a sample is a handful of sysfs reads, ~0.1 ms of host time —
        # inside the long block eight of them cost it 3 us per step)
        n_clk = int(min(96, max(16, np.ceil(0.5 / max(t_b, 1e-6)))))
        for i in range(n_clk):
            step()
            if rank == 0 and pci and i % max(n_clk // 8, 1) == max(n_clk // 8, 1) - 1 and len(psamples) < 8:
                psamples.append(device_power_state(pci))
        barrier()
        psamples = [q for q in psamples if isinstance(q, dict) and isinstance(q.get("sclk_mhz"), int)]
        if psamples:
            power_state = {"pci_bus_id": pci, "samples_while_stepping": len(psamples),
                           "sclk_mhz": {"min": min(q["sclk_mhz"] for q in psamples), "mean": float(np.mean([q["sclk_mhz"] for q in psamples])), "max": max(q["sclk_mhz"] for q in psamples)},
                           "mclk_mhz": psamples[-1].get("mclk_mhz"), "fclk_mhz": psamples[-1].get("fclk_mhz"), "power_cap_w": psamples[-1].get("power_cap_w"),
                           "power_w": {"mean": float(np.mean([q.get("power_w") or q.get("power_avg_w") or 0.0 for q in psamples])),
                                       "max": max(q.get("power_w") or q.get("power_avg_w") or 0.0 for q in psamples)}}
        else:
            power_state = {"pci_bus_id": pci, "all_cards_after_block": device_power_state()}
        m.kernel_timing_all(False)
        rows = [m.kernel_ms_all(sl) for sl in lslots]
        rows = [r for r in rows if min(r) >= 0]
        clocks = m.clock_probe()
        bracket_us = m.bracket_overhead_us(50)
        if rows:
            mean = [float(np.mean([r[k] for r in rows])) for k in range(4)]
            kern_all = {"prep_pose_texel": mean[0], "warp": mean[1], "post_warp_a": mean[2], "gram": mean[3], "samples": len(rows), "n_long": n_long}

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
        wk_raw = float(np.mean(warp_ms))
        # The bracket's own cost, calibrated in this run, two ways.  (i) empty_kernel_bracket_us: an EMPTY kernel between two events, queued behind work —
        # an upper bound (it contains the empty kernel's own dispatch-to-completion time, which a profiler also counts as that kernel's duration).
        # (ii) bracket_overhead_us: the steps sampled with consecutive events around all four launches are slower than their unsampled neighbours of the
        # same block by exactly what the event records add — (sum of the four intervals - ms_per_step_long) / 4 per interval.  The dominant kernel's
        # time is reported NET of (ii) (the figure rocprofv3's kernel trace agrees with), the raw reading beside it.
        bracket_cost_us = None
        if kern_all and long_ms:
            bracket_cost_us = max(0.0, (kern_all["prep_pose_texel"] + kern_all["warp"] + kern_all["post_warp_a"] + kern_all["gram"] - long_ms) * 1e3 / 4.0)
        wk = max(wk_raw - (bracket_cost_us if bracket_cost_us is not None else 0.0) * 1e-3, 1e-6)
        n_launch = local.size()
        n_cand = m.event_counts()[1]
        inl_frac = float(n_inl) / max(n_launch, 1)
        # inlier-weighted byte model (VERDICT r4 #4c): event + link + predecessor are read for every event (36 B); the map gathers, ep, the count
        # and A22 | b2 read-modify-writes and the 112-B A12 factor happen per INLIER (208 B)
        wbytes = 36.0 + 208.0 * inl_frac
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
            "ms_per_step_long": long_ms,
            "config": {"workload": w.describe(), "events_per_gpu": args.events_per_gpu, "total_events": n_total,
                       "thres_valid_pixel": w.thres_valid_pixel, "alpha": w.alpha, "cost": "quadratic",
                       "step": "evaluateDataError(eval_deriv)+formNormalEq+applyL2Reg, inputs resident in HBM; " +
                               ("the reference-order residual vector ep is produced in every step" if ep_in_step else
                                "residuals stay per event in HBM (the compacted ep vector is produced when it is asked for: with_ep_ms_per_step)"),
                       "parallelism": f"time-sharded x{world}" if world > 1 else (f"shard {args.shard_rank} of {args.shard_of} of the window, one GPU, no collectives" if args.shard_of > 1 else "single GPU"),
                       "events_per_rank": int(local.size()), "collectives_ms_per_step": coll_ms, "with_ep_ms_per_step": with_ep_ms, "no_ep_ms_per_step": no_ep_ms, "ep_in_step": ep_in_step,
                       "backend": ("gloo, all ranks on device 0 (rehearsal)" if args.one_device else "rccl") if use_dist else None,
                       "inliers_rank0": int(n_inl), "inlier_frac": inl_frac, "candidates_rank0": int(n_cand), "active_pixels": int(sh.P), "set_events_s": round(t_set, 3),
                       "settle": settle, "setup": m.setup_info()},
            "roofline": {"bound": "hbm", "kernel": "emba_warp_tiled_kernel" if m.setup_info()["tile_order"] else "emba_warp_residual_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": (tr["hbm_bytes_per_launch"] if tr else None), "traffic_unit": "bytes/launch (rocprofv3 PMC, profiles/)",
                         "counter_GBs": counter, "counter_frac": (counter / HBM_PEAK_GBS if counter else None),
                         "algorithmic_bytes_per_launch": ALGO_BYTES_PER_EVENT * n_launch,
                         "bytes_per_event": ALGO_BYTES_PER_EVENT, "events_per_launch": n_launch, "kernel_ms": wk, "kernel_ms_raw": wk_raw,
                         "bracket_overhead_us": bracket_cost_us, "empty_kernel_bracket_us": bracket_us,
                         "accumulate_kernel_ms": float(np.mean(accum_ms)),
                         "path_frac": ALGO_BYTES_PER_EVENT * n_launch / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "inlier_weighted": {"bytes_per_event": wbytes, "model": "36 + 208 x inlier_frac", "inlier_frac": inl_frac,
                                             "frac": wbytes * n_launch / (wk * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                             "path_frac": wbytes * n_launch / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS}},
        }
        if kern_all:
            ksum = kern_all["prep_pose_texel"] + kern_all["warp"] + kern_all["post_warp_a"] + kern_all["gram"]
            # consecutive events: the four intervals tile a sampled step from its first launch to the end of its last kernel (each includes the
            # record of the event that ends it, ~bracket_overhead_us less what queuing hides); gap_ms = what a step of the SAME block costs on
            # the host's clock beyond that — negative when the sampled steps' event records make them slower than the unsampled ones
            out["kernels_ms"] = dict(kern_all, sum=ksum, gap_ms=(long_ms - ksum) if long_ms else None,
                                     gap_vs_timed_region_ms=ms_per_step - ksum,
                                     note="HIP events around all launches of every 16th step of the long block; intervals, not net of the bracket overhead")
        out["device"] = {"clock_probe": clocks, "sysfs": power_state,
                         "note": "sysfs: the card's own report sampled WHILE a block of steps ran, right behind the long block (the clock the kernels had; at the power cap it sits below the "
                                 "attribute's peak).  clock_probe: shader cycles (s_memtime) per second of the constant-rate clock around a light fp32 chain on every "
                                 "SIMD, right behind the block — the clock the chip returns to when the load is light"}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(w)
        print(json.dumps(out), file=json_out, flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
