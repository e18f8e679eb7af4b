"""what the C++ resident host costs per LM iteration.  Writes a window (default: config 2's shape — 10 M events over 10 s, K = 201,
1024 x 2048) in tests/cpp's input layout, builds tests/cpp/resident_test.cpp (emba_host::solveTimeWindow, emba_amd/host/solve_time_window.hpp) and runs it
three times on the same context (the first window also allocates the context's buffers), on one rank and on two ranks of one device; the same window
through the Python resident host (emba_amd.solver) for comparison.
    python scripts/resident_host_timing.py [n_events pano_h K dt_knots [max_iter]]"""
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
with tempfile.TemporaryDirectory() as d:
    exe = os.path.join(d, "resident_test")
    subprocess.check_call(["g++", "-std=c++17", "-O2", os.path.join(ROOT, "tests", "cpp", "resident_test.cpp"), "-o", exe, "-L", os.path.join(ROOT, "emba_amd"), "-lemba_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "emba_amd"), "-lpthread"])
    p = os.path.join(d, "in.bin")
    with open(p, "wb") as f:
        f.write(struct.pack("<6i", w.sensor_w, w.sensor_h, w.pano_w, w.pano_h, w.K, w.thres_valid_pixel))
        f.write(struct.pack("<3q", w.traj.t0_ns, w.traj.dt_ns, w.events.size()))
        f.write(struct.pack("<2d", w.C_th, w.alpha))
        for arr, ty in ((w.lut, "<f8"), (init.knots_xyzw, "<f8"), (w.Gx, "<f8"), (w.Gy, "<f8"), (w.events.x, "<u2"), (w.events.y, "<u2"), (w.events.polarity, "u1"), (w.events.t_ns, "<i8")):
            f.write(np.ascontiguousarray(arr).astype(ty).tobytes())
    if os.environ.get("PROF_DEVICES"):      # PROF_DEVICES=0,0: rocprofv3 kernel stats of one window on those ranks instead of the timing runs (what do two ranks on one device spend?)
        import glob
        out = os.path.join(ROOT, "gpurun_out", "prof_resident_" + os.environ["PROF_DEVICES"].replace(",", "_"))
        subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", out, "--", exe, p, os.environ["PROF_DEVICES"], str(max_iter), "0", "0", "", "1"],
                       capture_output=True, text=True, timeout=900, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"))
        for f in glob.glob(out + "/*/*_kernel_stats.csv"):
            print(f"kernel stats of one window, devices {os.environ['PROF_DEVICES']}, {max_iter} LM iterations:")
            for l in open(f).read().splitlines()[:28]:
                c = l.split('","')
                print("   " + " | ".join(x.strip('"')[:70] for x in c[:4]))
        sys.exit(0)
    for devs in ("0", "0,0"):
        r = subprocess.run([exe, p, devs, str(max_iter), "0", "0", "", "3"], capture_output=True, text=True, timeout=900)
        win = [l for l in r.stdout.splitlines() if l.startswith("WINDOW ")]
        acc = [int(l.split()[5]) for l in r.stdout.splitlines() if l.startswith("LM ")]
        if r.returncode != 0 or not win:
            print(f"resident_test failed on devices {devs}: rc {r.returncode}\n{r.stdout[-600:]}{r.stderr[-600:]}"); continue
        print(f"C++ resident host (emba_host::solveTimeWindow), N={n} K={K} pano {ph}x{2*ph}, devices {devs}; accepted {sum(acc)} of {len(acc)}:")
        for l in win:
            print("   " + l)
m = LEGM(w.sensor_w, w.sensor_h, w.lut, w.C_th, w.pano_w, w.pano_h)
for rep in range(2):
    t0 = time.perf_counter()
    r = solve_time_window(m, init, w.events, w.Gx, w.Gy, BASettings(), LMSettings(max_num_iter=max_iter), resident=True)
    wall = time.perf_counter() - t0
    print(f"Python resident host (emba_amd.solver), same window, pass {rep}: {r.iterations} iterations in {wall*1e3:.1f} ms all in -> {wall/max(r.iterations,1)*1e3:.2f} ms per iteration")
