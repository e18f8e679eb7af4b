import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from emba_amd import LEGM
from emba_amd.synth import make_workload
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1200000
use_torch_stream = len(sys.argv) > 2 and sys.argv[2] == "torch"
w = make_workload(n_events=n)
stream = None
if use_torch_stream:
    ts_ = torch.cuda.Stream(); torch.cuda.set_stream(ts_); stream = ts_.cuda_stream
m = LEGM(w.sensor_w, w.sensor_h, w.lut, w.C_th, w.pano_w, w.pano_h, stream=stream)
m.set_events(w.events); m.upload_map(w.Gx, w.Gy)
m.sync()
ts = []
for i in range(4000):
    t0 = time.perf_counter(); m.step(w.traj, w.thres_valid_pixel, w.alpha); ts.append(time.perf_counter() - t0)
m.sync()
ts = np.array(ts) * 1e6
print(n, "torch stream" if use_torch_stream else "own stream", "median %.1f us  mean %.1f us; steps > 1 ms:" % (np.median(ts), ts.mean()), [(int(i), round(float(ts[i]))) for i in np.nonzero(ts > 1000)[0]])
