"""f3 timing: emba_reconstruct_intensity on the resident map (no host copies in the timed region) vs the numpy/scipy oracle."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from emba_amd import LEGM
from emba_amd.synth import pinhole_bearing_lut
from oracle import poisson as OP

for H in (512, 1024, 2048):
    W = 2 * H
    m = LEGM(64, 48, pinhole_bearing_lut(64, 48, 60., 60., 32., 24.), 0.2, W, H)
    rng = np.random.default_rng(0)
    Gx, Gy = rng.normal(size=(H, W)), rng.normal(size=(H, W))
    m.upload_map(Gx, Gy)
    m.reconstructIntensity(download=False)          # builds the sine matrices
    m.timer_start(0)
    for _ in range(5):
        m.reconstructIntensity(download=False)
    m.timer_stop(0)
    ms = m.timer_ms(0) / 5
    flop = 2 * (2.0 * H * W * W + 2.0 * H * H * W)
    t0 = time.time(); OP.reconstruct_from_gradient(Gx, Gy); cpu = time.time() - t0
    print(f"{H}x{W}: device {ms:.3f} ms = {flop / ms / 1e9:.1f} TFLOP/s fp64 ({flop / ms / 1e9 / 78.6:.2f} of the 78.6 TFLOP/s matrix peak); scipy pocketfft on the host {cpu * 1e3:.1f} ms")
