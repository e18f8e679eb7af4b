"""f3 timing: emba_reconstruct_intensity on the resident map (no host copies in the timed region), default path (Fourier analysis along H + tridiagonal
solves along W) and, with `dense` as argument, option poisson = 1 (four sine-matrix GEMMs), with the error against the numpy/scipy oracle."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emba_amd import LEGM
from emba_amd.synth import pinhole_bearing_lut
from oracle import poisson as OP

mode = sys.argv[1] if len(sys.argv) > 1 else "default"
for H in (75, 512, 1024, 2048):
    W = 2 * H
    m = LEGM(64, 48, pinhole_bearing_lut(64, 48, 60., 60., 32., 24.), 0.2, W, H)
    m.set_option("poisson", {"default": 0, "dense": 1, "nofold": 2}[mode])
    rng = np.random.default_rng(0)
    Gx, Gy = rng.normal(size=(H, W)), rng.normal(size=(H, W))
    m.upload_map(Gx, Gy)
    M = m.reconstructIntensity()          # builds the tables
    m.timer_start(0)
    for _ in range(5):
        m.reconstructIntensity(download=False)
    m.timer_stop(0)
    ms = m.timer_ms(0) / 5
    t0 = time.time(); Mo = OP.reconstruct_from_gradient(Gx, Gy); cpu = time.time() - t0
    err = np.abs(M - Mo).max() / np.abs(Mo).max()
    print(f"{mode:8s} {H}x{W}: device {ms:.3f} ms; max rel err vs oracle {err:.2e}; scipy pocketfft on the host {cpu * 1e3:.1f} ms")
    m.close()
