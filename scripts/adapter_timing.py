"""what the drop-in costs per LM iteration.  Writes a window (default: config 2's shape — 10 M events over 10 s, K = 201, 1024 x 2048) in
tests/cpp/adapter_test's input layout, runs the prebuilt binary (the EMBA::LEGM adapter driven through solver.cpp's call order) with ADAPTER_TIMING=1 on
one rank and on EMBA_HIP_DEVICES=0,0, and the same window through the resident Python host (emba_amd.solver) for comparison.
    python scripts/adapter_timing.py [n_events pano_h K dt_knots [max_iter]]"""
import os, struct, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from emba_amd import LEGM
from emba_amd.synth import make_workload
from emba_amd.solver import BASettings, LMSettings, solve_time_window
from test_lm_solver_cpu import perturbed
a = sys.argv[1:]
n, ph, K, dt = (int(a[0]), int(a[1]), int(a[2]), float(a[3])) if len(a) >= 4 else (10_000_000, 1024, 201, 0.05)
max_iter = int(a[4]) if len(a) >= 5 else 6
w = make_workload(n_events=n, pano_h=ph, K=K, dt_knots=dt)
init = perturbed(w, 0.002)
exe = os.path.join(ROOT, "tests", "cpp", "_build", "adapter_test")
with tempfile.TemporaryDirectory() as d:
    p = os.path.join(d, "in.bin")
    with open(p, "wb") as f:
        f.write(struct.pack("<6i", w.sensor_w, w.sensor_h, w.pano_w, w.pano_h, w.K, w.thres_valid_pixel))
        f.write(struct.pack("<3q", w.traj.t0_ns, w.traj.dt_ns, w.events.size()))
        f.write(struct.pack("<2d", w.C_th, w.alpha))
        for arr, ty in ((w.lut, "<f8"), (init.knots_xyzw, "<f8"), (w.Gx, "<f8"), (w.Gy, "<f8"), (w.events.x, "<u2"), (w.events.y, "<u2"), (w.events.polarity, "u1"), (w.events.t_ns, "<i8")):
            f.write(np.ascontiguousarray(arr).astype(ty).tobytes())
    for devs in os.environ.get("ADAPTER_DEVICES", "0;0,0").split(";"):      # e.g. ADAPTER_DEVICES="0,0,0,0,0,0,0,0" (eight ranks on one device)
        env = dict(os.environ); env["EMBA_HIP_DEVICES"] = devs; env["ADAPTER_TIMING"] = "1"
        r = subprocess.run([exe, p, str(max_iter), "0", "0"], capture_output=True, text=True, timeout=600, env=env)
        ts = [float(l.split()[2]) for l in r.stdout.splitlines() if l.startswith("TIME ")]
        acc = [int(l.split()[5]) for l in r.stdout.splitlines() if l.startswith("LM ")]
        if r.returncode != 0 or not ts:
            print(f"adapter_test failed on devices {devs}: rc {r.returncode}\n{r.stdout[-600:]}{r.stderr[-600:]}"); continue
        # (the first iteration includes the window's registration, order preparation and first full upload: reported apart)
        print(f"drop-in (EMBA::LEGM adapter, solver.cpp's call order), N={n} K={K} pano {ph}x{2*ph}, devices {devs}: first iteration {ts[0]:.1f} ms, "
              f"following {len(ts)-1} iterations mean {np.mean(ts[1:]):.2f} ms (min {np.min(ts[1:]):.2f}, max {np.max(ts[1:]):.2f}); accepted {sum(acc)} of {len(acc)}")
        for l in r.stdout.splitlines():
            if l.startswith("PHASES "):
                print("      " + l)
        if os.environ.get("EMBA_ADAPTER_TRACE"):
            for l in r.stderr.splitlines():
                if l.startswith("[adapter]") or l.startswith("[group ep]"):
                    print("      " + l)
        if "solve_debug" in os.environ.get("EMBA_HIP_OPTIONS", ""):      # EMBA_HIP_OPTIONS=solve_debug=1: the stages of every sharded solve (stderr of the library)
            for l in r.stderr.splitlines():
                if l.startswith("[group solve]"):
                    print("      " + l)
m = LEGM(w.sensor_w, w.sensor_h, w.lut, w.C_th, w.pano_w, w.pano_h)
t0 = time.perf_counter(); m.set_events(w.events); m.upload_map(w.Gx, w.Gy); t_set = time.perf_counter() - t0
t0 = time.perf_counter()
r = solve_time_window(m, init, w.events, w.Gx, w.Gy, BASettings(), LMSettings(max_num_iter=max_iter), resident=True)
wall = time.perf_counter() - t0
print(f"resident host (emba_amd.solver), same window: {r.iterations} iterations in {wall*1e3:.1f} ms incl. set_events + upload ({t_set*1e3:.1f} ms measured apart) "
      f"-> {(wall - t_set)/max(r.iterations,1)*1e3:.2f} ms per iteration")
