#!/usr/bin/env python3
"""Event-based mosaicing bundle adjustment of one time window on an MI355X, without ROS: the job of the reference's
`roslaunch emba <seq>.launch` back-end (src/emba/emba.cpp:29-330 + solver.cpp:11-368) for data already on disk.

  python examples/run_ba.py --demo out/                       # simulate a scene, perturb the trajectory, refine, write results
  python examples/run_ba.py --events ev.npz --poses init_traj.txt --map-dir init_map/ --calib calib.npz --out out/

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
from emba_amd.solver import BASettings, LMSettings, solve_time_window  # noqa: E402


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
    model = LEGM(sw, sh, lut, C_th, W, H)
    ba = BASettings(use_IRLS=a.cost != "quadratic", cost_type=a.cost, eta=a.eta, thres_valid_pixel=a.thres_valid_pixel, alpha=a.alpha,
                    damping_factor=a.damping_factor)
    print(f"{events.size()} events, {traj.size()} control poses, panorama {H}x{W}")
    t0 = time.time()
    res = solve_time_window(model, traj, events, Gx, Gy, ba, LMSettings(max_num_iter=a.max_iter), verbose=a.verbose, resident=True)
    dt = time.time() - t0
    print(f"{res.iterations} LM iterations in {dt * 1e3:.1f} ms ({'converged' if res.converged else 'stopped'}), cost {res.cost_min:.6e}")
    if truth is not None:
        err = lambda tr: np.degrees(np.mean([np.linalg.norm(so3.log(so3.mul(so3.inverse(p), q))) for p, q in zip(tr.knots_xyzw, truth.knots_xyzw)]))
        print(f"mean control-pose error vs ground truth: {err(traj):.4f} deg -> {err(res.traj):.4f} deg")
    eio.write_trajectory(os.path.join(a.out, "refined_traj.txt"), res.traj)
    eio.save_map(a.out, *model.downloadMap())
    # intensity panorama from the refined gradient map (solver.cpp:417-425 / 471-479), reconstructed on the device
    eio.save_pgm(os.path.join(a.out, "map_poisson_opt.pgm"), eio.normalize_robust(model.reconstructIntensity(), 0.1))


if __name__ == "__main__":
    main()
