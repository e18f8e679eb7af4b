import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from emba_amd import LEGM
from emba_amd.synth import make_workload
n, sensor, pano_h, K = 100_000_000, (240, 180), 2048, 256
w = make_workload(n_events=1000, pano_h=pano_h, K=K, sensor=sensor)
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
for rep in range(2):
    m.set_events_dev(x.data_ptr(), y.data_ptr(), pol.data_ptr(), t.data_ptr(), n)
    m.eval_launch(w.traj); m.eval_finish(); m.sync()
    print(m.setup_info(), flush=True)
