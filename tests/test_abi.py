"""CPU tests of the boundary: the C-ABI library builds, loads and exports every symbol include/emba_hip.h declares;
without a GPU it fails loudly (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "emba_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(emba_[a-zA-Z0-9_]+)\s*\(", src)))


def test_header_symbols_all_exported_and_bound(hip_lib):
    from emba_amd import _lib
    syms = _declared_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(hip_lib, s), f"{s} declared in include/emba_hip.h but not exported by libemba_hip.so"
    assert sorted(_lib.SIGNATURES) == syms, "ctypes binding table and header disagree"
    assert hip_lib.emba_abi_version() == 2
    assert b"gfx950" in hip_lib.emba_build_info()


def test_library_contains_gfx950_code_objects(hip_lib):
    from emba_amd import _lib
    blob = open(_lib.LIB_PATH, "rb").read()
    assert b"gfx950" in blob and b"emba_warp_residual_kernel" in blob and b"emba_gram_kernel" in blob


def test_no_cpu_fallback_without_gpu(hip_lib):
    """emba_create must refuse to run when no HIP device exists; nothing in emba_amd/ may import the oracle."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from emba_amd import LEGM, EmbaError
    from emba_amd.synth import pinhole_bearing_lut
    with pytest.raises(EmbaError) as ei:
        LEGM(8, 8, pinhole_bearing_lut(8, 8, 10, 10, 4, 4), 0.2, 64, 32)
    assert ei.value.status == 2  # EMBA_ERR_NO_DEVICE
    for dirpath, _, files in os.walk(os.path.join(ROOT, "emba_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "import oracle" not in txt and "from oracle" not in txt and "libemba_oracle" not in txt, f


def test_create_argument_validation(hip_lib):
    from emba_amd._lib import EmbaCfg
    ctx = C.c_void_p()
    assert hip_lib.emba_create(None, C.byref(ctx)) == 1
    lut = np.zeros(3 * 16)
    cfg = EmbaCfg(4, 4, 64, 32, lut.ctypes.data_as(C.POINTER(C.c_double)), 0.2, 50, 10.0, 0, None)
    assert hip_lib.emba_create(C.byref(cfg), C.byref(ctx)) == 1        # event_batch must be 100 (model.cpp:78)
    assert b"100" in hip_lib.emba_last_error(None)
    cfg = EmbaCfg(0, 4, 64, 32, lut.ctypes.data_as(C.POINTER(C.c_double)), 0.2, 100, 10.0, 0, None)
    assert hip_lib.emba_create(C.byref(cfg), C.byref(ctx)) == 1


def test_synthetic_workload_is_deterministic():
    from emba_amd.synth import make_workload
    a = make_workload(n_events=5000, pano_h=64, K=4, sensor=(16, 12), focal=12.0)
    b = make_workload(n_events=5000, pano_h=64, K=4, sensor=(16, 12), focal=12.0)
    assert np.array_equal(a.events.x, b.events.x) and np.array_equal(a.Gx, b.Gx) and np.array_equal(a.traj.knots_xyzw, b.traj.knots_xyzw)
    assert (np.diff(a.events.t_ns) > 0).all()
    assert a.events.t_ns[-1] < a.traj.t0_ns + a.traj.dt_ns * (a.K - 1)
    assert np.allclose(np.linalg.norm(a.traj.knots_xyzw, axis=1), 1.0, atol=1e-15)
