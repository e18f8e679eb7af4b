#!/usr/bin/env python3
"""Once-per-window cost on the device: emba_set_events_dev (validation, batch midpoints, (sensor pixel, time) radix sort) with the event
arrays already in HBM, and the order preparation of the first evaluation (control-pose pairs, record slots, tile order)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from emba_amd import LEGM
from emba_amd.synth import make_workload

for n, sensor, pano_h, K in ((1_000_000, (240, 180), 1024, 21), (10_000_000, (240, 180), 1024, 97), (40_000_000, (640, 480), 1024, 97), (100_000_000, (240, 180), 2048, 256)):
    w = make_workload(n_events=1000, pano_h=pano_h, K=K, sensor=sensor, focal=200.0 * sensor[0] / 240)     # map / LUT / trajectory only
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev); g.manual_seed(1)
    x = torch.randint(0, sensor[0], (n,), device=dev, generator=g, dtype=torch.int32).to(torch.int16)
    y = torch.randint(0, sensor[1], (n,), device=dev, generator=g, dtype=torch.int32).to(torch.int16)
    pol = torch.randint(0, 2, (n,), device=dev, generator=g, dtype=torch.int32).to(torch.uint8)
    T = w.traj.dt_ns * (K - 1)
    t = w.traj.t0_ns + (torch.arange(n, device=dev, dtype=torch.int64) * T) // n
    m = LEGM(w.sensor_w, w.sensor_h, w.lut, w.C_th, w.pano_w, w.pano_h, device=0)
    m.upload_map(w.Gx, w.Gy)
    torch.cuda.synchronize()
    best = None
    for rep in range(3):
        t0 = time.perf_counter()
        m.set_events_dev(x.data_ptr(), y.data_ptr(), pol.data_ptr(), t.data_ptr(), n)
        t1 = time.perf_counter()
        m.eval_launch(w.traj); m.eval_finish(); m.sync()
        info = m.setup_info()
        best = info if best is None or info["set_events_ms"] < best["set_events_ms"] else best
    print(f"N={n:>11,d} sensor {sensor[0]}x{sensor[1]} K={K:3d}: set_events_dev {best['set_events_ms']:8.2f} ms   first-evaluation order preparation {best['prepare_ms']:8.2f} ms "
          f"({'tile' if best['tile_order'] else 'pixel'} order, {best['entries']:,d} entries)", flush=True)
    m.close()
