#!/usr/bin/env python3
"""Event-based mosaicing bundle adjustment of one time window on an MI355X, without ROS: the job of the reference's
`roslaunch emba <seq>.launch` back-end (src/emba/emba.cpp:29-330 + solver.cpp:11-368) for data already on disk.

  python examples/run_ba.py --demo out/                       # simulate a scene, perturb the trajectory, refine, write results
  python examples/run_ba.py --events ev.npz --poses init_traj.txt --map-dir init_map/ --calib calib.npz --out out/
  python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/run_ba.py ... # the window's events time-sharded
                                                  # over the GPUs of one node (RCCL); every rank runs the same LM loop, rank 0 writes

Inputs: events (.npz: x, y u16; polarity u8; t_ns i64), initial poses ("t tx ty tz qx qy qz qw" per line), initial map
(Gx.bin / Gy.bin raw float64, H x 2H), calibration (.npz: K [3,3], D [<=5] plumb_bob, width, height).
Outputs: <out>/refined_traj.txt, <out>/Gx.bin, <out>/Gy.bin (the files emba.cpp:300-330 writes), <out>/map_poisson_opt.pgm."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from emba_amd import LEGM, io as eio, so3, synth                      # noqa: E402
from emba_amd.legm import LinearTrajectory                            # noqa: E402
from emba_amd.solver import BASettings, LMSettings, RuntimeLog, solve_time_window  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--demo", action="store_true")
    ap.add_argument("--events"); ap.add_argument("--poses"); ap.add_argument("--map-dir"); ap.add_argument("--calib")
    ap.add_argument("--dt-knots", type=float, default=0.05)
    ap.add_argument("--t-beg", type=float); ap.add_argument("--t-end", type=float)
    ap.add_argument("--C-th", type=float, default=0.2)
    ap.add_argument("--alpha", type=float, help="map L2 weight (default 5.0; 0 in --demo, where the initial map is already the true one)")
    ap.add_argument("--thres-valid-pixel", type=int, default=5)
    ap.add_argument("--damping-factor", type=float, default=1.0)
    ap.add_argument("--cost", default="quadratic", choices=["quadratic", "huber", "cauchy"])
    ap.add_argument("--eta", type=float, default=0.1)
    ap.add_argument("--max-iter", type=int, default=50)
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--runtime-log", action="store_true", help="write the reference's run-time records under <out>/final_results (every timed phase then ends in a host synchronisation)")
    ap.add_argument("--sharded", action="store_true", help="go through the multi-GPU host (ShardedLEGM / ShardedModel) even with one rank")
    a = ap.parse_args()
    if a.alpha is None:
        a.alpha = 0.0 if a.demo else 5.0
    os.makedirs(a.out, exist_ok=True)

    if a.demo:
        w = synth.make_scene_workload(pano_h=512, K=11, sensor=(128, 96), focal=120.0, n_steps=2000)
        rng = np.random.default_rng(5)
        knots = w.traj.knots_xyzw.copy()
        for i in range(1, len(knots)):
            knots[i] = so3.mul(so3.exp(rng.normal(size=3) * 0.01), knots[i])
        traj, truth = LinearTrajectory(knots, w.traj.t0_ns, w.traj.dt_ns), w.traj
        events, Gx, Gy, lut, sw, sh, C_th = w.events, w.Gx, w.Gy, w.lut, w.sensor_w, w.sensor_h, w.C_th
    else:
        cal = np.load(a.calib)
        sw, sh = int(cal["width"]), int(cal["height"])
        lut = eio.bearing_lut_from_calibration(cal["K"], cal["D"], sw, sh)
        Gx, Gy = eio.load_map(a.map_dir)
        t, qs = eio.load_poses(a.poses)
        t_beg = a.t_beg if a.t_beg is not None else t[0]
        t_end = a.t_end if a.t_end is not None else t[-1]
        num_cps = int(round((t_end - t_beg) / a.dt_knots)) + 1                      # trajectory.cpp:231-245
        sel = (t >= t_beg) & (t <= t_end)
        traj = LinearTrajectory.from_seconds(t_beg, a.dt_knots, eio.fit_ctrl_poses(t[sel], qs[sel], t_beg, a.dt_knots, num_cps))
        events = eio.load_events(a.events, int(t_beg * 1e9), traj.t0_ns + traj.dt_ns * (num_cps - 1) - 1)
        truth, C_th = None, a.C_th

    H, W = Gx.shape
    world, rank, local_rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    legm = None
    if world > 1 or a.sharded:
        # one process per GPU (torchrun): kernels and RCCL collectives share one explicit torch stream
        import torch
        import torch.distributed as dist
        from emba_amd.sharded import HipEngine, ShardedLEGM, ShardedModel
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29519")
        dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
        tstream = torch.cuda.Stream(device=dev)
        torch.cuda.set_stream(tstream)
        legm = LEGM(sw, sh, lut, C_th, W, H, device=local_rank, stream=tstream.cuda_stream)
        count_t = torch.zeros(H * W, dtype=torch.int32, device=dev)
        pack_t = torch.zeros(9 * traj.size() ** 2 + 3 * traj.size() + 5 * H * W, dtype=torch.float64, device=dev)
        sh_ = ShardedLEGM(HipEngine(legm), dist, count_t, pack_t, sw, torch.zeros(H * W, dtype=torch.uint8, device=dev))
        sh_.force_collectives = world == 1
        model = ShardedModel(sh_, legm)
    else:
        model = legm = LEGM(sw, sh, lut, C_th, W, H)
    ba = BASettings(use_IRLS=a.cost != "quadratic", cost_type=a.cost, eta=a.eta, thres_valid_pixel=a.thres_valid_pixel, alpha=a.alpha,
                    damping_factor=a.damping_factor)
    if rank == 0:
        print(f"{events.size()} events, {traj.size()} control poses, panorama {H}x{W}" + (f", {world} rank(s) through the sharded host" if legm is not model else ""))
    t0 = time.time()
    # the reference's run-time records (final_results/runtime_{formEqs,solveEqs,objFuncs}.txt, iterations.txt: solver.cpp:105-151, 170-178, 205-223, 271-291)
    rlog = RuntimeLog(a.out) if (a.runtime_log and rank == 0) else None
    res = solve_time_window(model, traj, events, Gx, Gy, ba, LMSettings(max_num_iter=a.max_iter), verbose=a.verbose, resident=True, runtime_log=rlog)
    dt = time.time() - t0
    if rank != 0:                                           # every rank holds the same result; rank 0 writes it
        import torch.distributed as dist
        dist.barrier(); dist.destroy_process_group()
        return
    print(f"{res.iterations} LM iterations in {dt * 1e3:.1f} ms ({'converged' if res.converged else 'stopped'}), cost {res.cost_min:.6e}")
    if truth is not None:
        err = lambda tr: np.degrees(np.mean([np.linalg.norm(so3.log(so3.mul(so3.inverse(p), q))) for p, q in zip(tr.knots_xyzw, truth.knots_xyzw)]))
        print(f"mean control-pose error vs ground truth: {err(traj):.4f} deg -> {err(res.traj):.4f} deg")
    eio.write_trajectory(os.path.join(a.out, "refined_traj.txt"), res.traj)
    eio.save_map(a.out, *model.downloadMap())
    # intensity panorama from the refined gradient map (solver.cpp:417-425 / 471-479), reconstructed on the device
    eio.save_pgm(os.path.join(a.out, "map_poisson_opt.pgm"), eio.normalize_robust(legm.reconstructIntensity(), 0.1))
    if legm is not model:
        import torch.distributed as dist
        dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    main()
