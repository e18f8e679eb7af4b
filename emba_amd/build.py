"""In-tree build of the HIP extension: hipcc cross-compiles gfx950 code objects without a GPU."""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
SRC = os.path.join(_HERE, "csrc", "emba_hip.hip")
import glob
DEPS = [SRC, os.path.join(ROOT, "include", "emba_hip.h")] + sorted(glob.glob(os.path.join(_HERE, "csrc", "*.h")))   # every header the .hip includes
OUT = os.path.join(_HERE, "libemba_hip.so")

HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-munsafe-fp-atomics"]


def hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def build_hip(force=False, verbose=False):
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in DEPS):
        return OUT
    cmd = [hipcc()] + HIPCC_FLAGS + [SRC, "-o", OUT]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=ROOT)
    return OUT
