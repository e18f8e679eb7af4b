"""The C++ host side above the C ABI (emba_amd/host/legm_host.hpp — mirror of EMBA::LEGM on plain containers):
compiles and links on CPU; on the GPU it runs evaluateDataError/formNormalEq/applyL2Reg and is checked against oracle values."""
import os
import struct
import subprocess

import numpy as np
import pytest

from helpers import oracle_run, small_workload

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path, hip_lib, name="host_test"):
    exe = str(tmp_path / name)
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-Werror", os.path.join(ROOT, "tests", "cpp", name + ".cpp"), "-o", exe,
                           "-L", os.path.join(ROOT, "emba_amd"), "-lemba_hip", "-Wl,-rpath," + os.path.join(ROOT, "emba_amd")])
    return exe


def _write_case(path, w, o, solve=None):
    ne = o["ne"]
    with open(path, "wb") as f:
        f.write(struct.pack("<6i", w.sensor_w, w.sensor_h, w.pano_w, w.pano_h, w.K, w.thres_valid_pixel))
        f.write(struct.pack("<3q", w.traj.t0_ns, w.traj.dt_ns, w.events.size()))
        f.write(struct.pack("<2d", w.C_th, w.alpha))
        for a, dt in ((w.lut, "<f8"), (w.traj.knots_xyzw, "<f8"), (w.Gx, "<f8"), (w.Gy, "<f8"), (w.events.x, "<u2"), (w.events.y, "<u2"),
                      (w.events.polarity, "u1"), (w.events.t_ns, "<i8")):
            f.write(np.ascontiguousarray(a).astype(dt).tobytes())
        f.write(struct.pack("<q", o["ep"].size)); f.write(o["ep"].astype("<f8").tobytes()); f.write(o["num_ev_map"].astype("<i4").tobytes())
        f.write(struct.pack("<q", ne["P"]))
        f.write(np.asfortranarray(ne["A11"]).ravel(order="F").astype("<f8").tobytes()); f.write(ne["b1"].astype("<f8").tobytes())
        f.write(ne["A22"].astype("<f8").tobytes()); f.write(ne["b2"].astype("<f8").tobytes()); f.write(ne["active"].astype("<u4").tobytes())
        if solve is not None:
            lam, fix, x1, x2 = solve
            f.write(struct.pack("<d", lam)); f.write(struct.pack("<i", 1 if fix else 0))
            f.write(x1.astype("<f8").tobytes()); f.write(x2.astype("<f8").tobytes())


def test_cpp_host_compiles_and_links(tmp_path, hip_lib):
    exe = _build(tmp_path, hip_lib)
    assert os.path.exists(exe)
    assert subprocess.run([exe], capture_output=True).returncode == 2      # usage error, but the loader resolved libemba_hip.so


@pytest.mark.gpu
def test_cpp_host_matches_oracle(tmp_path, hip_lib, oracle_mod):
    exe = _build(tmp_path, hip_lib)
    w = small_workload(n_events=20000)
    o = oracle_run(oracle_mod, w)
    p = tmp_path / "in.bin"
    _write_case(p, w, o)
    r = subprocess.run([exe, str(p)], capture_output=True, text=True, timeout=120)
    print(r.stdout, r.stderr)
    assert r.returncode == 0 and r.stdout.startswith("OK"), r.stdout + r.stderr


def test_cpp_sharded_host_compiles_and_links(tmp_path, hip_lib):
    exe = _build(tmp_path, hip_lib, "sharded_test")
    assert subprocess.run([exe], capture_output=True).returncode == 2


@pytest.mark.gpu
@pytest.mark.parametrize("devices,force_rccl", [("0", False), ("0,0", False), ("0,0,0", False), ("0", True)])
def test_cpp_sharded_host_matches_oracle(tmp_path, hip_lib, oracle_mod, devices, force_rccl):
    """The single-process multi-GPU host (emba_host::ShardedLEGM over emba_group_*): 1, 2 and 3 ranks on one GPU (two / three contexts and
    streams, in-library exchange), the whole LM-iteration call order incl. the sharded Schur solve, against the single-process oracle."""
    exe = _build(tmp_path, hip_lib, "sharded_test")
    w = small_workload(n_events=30000)
    o = oracle_run(oracle_mod, w, dense_A12=True)
    lam, fix = 1e-3, True
    x1, x2 = oracle_mod.solve_normal_eq(o["ne"], lam, fix)
    p = tmp_path / "in.bin"
    _write_case(p, w, o, (lam, fix, x1, x2))
    x2_split = "0" if devices == "0,0,0" else "1"     # option x2_split: exchange 2 split (rows on the side streams, under the Gram kernel) or in one piece
    # force_rccl (EMBA_GROUP_FORCE_RCCL): one rank through RCCL itself — run-time binding of librccl, ncclCommInitAll, grouped all-reduce / send / recv calls
    r = subprocess.run([exe, str(p), devices, x2_split, "1" if force_rccl else "0"], capture_output=True, text=True, timeout=180)
    print(r.stdout, r.stderr)
    last = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ""      # (RCCL prints a version banner first)
    assert r.returncode == 0 and last.startswith("OK"), r.stdout + r.stderr
    assert f"world={len(devices.split(','))} rccl={int(force_rccl)}" in last


ADAPTER_EXE = os.path.join(ROOT, "tests", "cpp", "_build", "adapter_test")


def test_adapter_binary_builds_where_the_reference_is_present(hip_lib):
    """The drop-in EMBA::LEGM (legm_adapter.hpp: all eight methods of model.h:76-128) compiles and LINKS against the mock of the reference's
    declarations + the reference's own Eigen (built here; the binary travels to the GPU box)."""
    if not os.path.isdir("/root/reference/thirdparty/basalt-headers/thirdparty/eigen"):
        pytest.skip("the reference's vendored Eigen is not present on this machine")
    import __graft_entry__ as ge
    exe = ge.build_adapter_test()
    assert exe and os.path.exists(exe)
    assert subprocess.run([exe], capture_output=True).returncode == 2


@pytest.mark.gpu
@pytest.mark.parametrize("devices,use_irls,use_cg", [("0", False, False), ("0", True, False), ("0", False, True), ("0,0,0", False, False), ("0,0", True, False), ("0,0", False, True)])
def test_adapter_runs_the_reference_call_order(tmp_path, hip_lib, oracle_mod, devices, use_irls, use_cg):
    """solveTimeWindow's call order (solver.cpp:63-353) through the EMBA::LEGM adapter — evaluateDataError, formNormalEq[IRLS], applyL2Reg,
    the first-window trim, solveNormalEq[CG], updateMap, evaluateDataError on the trial Mats, accept / reject inferred from the calls — on one
    rank and on 2 / 3 ranks of one GPU, against the same loop on the CPU oracle: decisions, costs, trajectory, map."""
    if not os.path.exists(ADAPTER_EXE):
        pytest.skip("tests/cpp/_build/adapter_test was not built (needs the reference's Eigen at build time)")
    from emba_amd import synth
    from emba_amd.solver import BASettings, LMSettings, solve_time_window
    from helpers import OracleModel
    from test_lm_solver_cpu import perturbed
    w = synth.make_scene_workload(n_steps=1000)
    init = perturbed(w)
    wi = synth.Workload(w.sensor_w, w.sensor_h, w.pano_w, w.pano_h, w.lut, w.C_th, w.Gx, w.Gy, init, w.events, w.thres_valid_pixel, 5.0)
    p = tmp_path / "in.bin"
    o = dict(ep=np.zeros(0), num_ev_map=np.zeros((w.pano_h, w.pano_w), np.int32),
             ne=dict(P=0, A11=np.zeros((3 * w.K, 3 * w.K)), b1=np.zeros(3 * w.K), A22=np.zeros((0, 2, 2)), b2=np.zeros(0), active=np.zeros(0, np.uint32)))
    _write_case(p, wi, o)           # (only the inputs are read by adapter_test)
    max_iter = 8
    env = dict(os.environ); env["EMBA_HIP_DEVICES"] = devices
    if devices == "0,0" and use_irls:
        # formNormalEqIRLS declares its cost AFTER the first evaluation — the gathered (quadratic) A22 | b2 rows are then NOT final and
        # must not travel through the split exchange 2 while emba_form_accumulate rebuilds them from the records: forced on here (it only switches
        # itself on from 3 M events per rank)
        env["EMBA_HIP_OPTIONS"] = "x2_split=1"
    r = subprocess.run([ADAPTER_EXE, str(p), str(max_iter), "1" if use_irls else "0", "1" if use_cg else "0"], capture_output=True, text=True, timeout=300, env=env)
    print(r.stdout[-3000:], r.stderr[-2000:])
    assert r.returncode == 0, r.stdout + r.stderr
    lines = r.stdout.strip().splitlines()
    lm = [l.split() for l in lines if l.startswith("LM ")]
    end = [l.split() for l in lines if l.startswith("END ")][0]
    knots = np.array([[float(v) for v in l.split()[1:]] for l in lines if l.startswith("KNOT ")])
    ba = BASettings(use_IRLS=use_irls, cost_type="huber", eta=0.1, alpha=5.0, use_CG=use_cg)
    om = OracleModel(oracle_mod, w, use_cg=use_cg)
    ro = solve_time_window(om, init, w.events, w.Gx, w.Gy, ba, LMSettings(max_num_iter=max_iter))
    assert len(lm) == len(ro.log) and int(end[1]) == ro.iterations and bool(int(end[2])) == ro.converged
    assert [int(l[5]) for l in lm] == [int(e[4]) for e in ro.log], "accept / reject sequence differs from the oracle loop"
    assert any(int(l[5]) for l in lm) and not all(int(l[5]) for l in lm), "the case is meant to contain accepted AND rejected steps"
    # CG stops at a relative residual of 1e-6 (model.cpp:823-824) and the device's summation order moves the stopping iteration by one or
    # two: each LM step's iterate is only that well defined and the difference carries into the following steps — decisions are compared
    # exactly, costs loosely
    tol = 1e-2 if use_cg else 1e-7
    for l, e in zip(lm, ro.log):
        assert float(l[3]) == pytest.approx(e[2], rel=tol) and float(l[4]) == pytest.approx(e[3], rel=tol)
    assert np.abs(knots - ro.traj.knots_xyzw).max() < (1e-3 if use_cg else 1e-7)
    Gx_o, Gy_o = om.downloadMap()
    n = Gx_o.size
    sx = float((Gx_o.ravel() * ((np.arange(n) % 7) + 1)).sum()); sy = float((Gy_o.ravel() * ((np.arange(n) % 5) + 1)).sum())
    mp_ = [l.split() for l in lines if l.startswith("MAP ")][0]
    assert float(mp_[1]) == pytest.approx(sx, rel=5e-2 if use_cg else 1e-7, abs=1e-9) and float(mp_[2]) == pytest.approx(sy, rel=5e-2 if use_cg else 1e-7, abs=1e-9)


def test_cpp_resident_host_compiles_and_links(tmp_path, hip_lib):
    exe = _build(tmp_path, hip_lib, "resident_test")
    assert subprocess.run([exe], capture_output=True).returncode == 2


@pytest.mark.gpu
@pytest.mark.parametrize("devices,use_irls,use_cg", [("0", False, False), ("0", True, False), ("0", False, True), ("0,0", False, False), ("0,0,0", True, False), ("0,0,0", False, True)])
def test_cpp_resident_solve_time_window(tmp_path, hip_lib, oracle_mod, devices, use_irls, use_cg):
    """EMBA::solveTimeWindow (solver.cpp:11-368) as a C++ host with everything of an iteration resident in HBM
    (emba_amd/host/solve_time_window.hpp on emba_host::ShardedLEGM; 1, 2 and 3 ranks on one GPU; Schur, Huber IRLS, CG) — against the same loop in
    emba_amd/solver.py on the device path (the log must agree decision for decision and cost for cost) and against the loop on the CPU oracle;
    the run-time records are written in the reference's line formats."""
    from emba_amd import LEGM, synth
    from emba_amd.solver import BASettings, LMSettings, solve_time_window
    from helpers import OracleModel
    from test_lm_solver_cpu import perturbed
    exe = _build(tmp_path, hip_lib, "resident_test")
    w = synth.make_scene_workload(n_steps=1000)
    init = perturbed(w)
    wi = synth.Workload(w.sensor_w, w.sensor_h, w.pano_w, w.pano_h, w.lut, w.C_th, w.Gx, w.Gy, init, w.events, w.thres_valid_pixel, 5.0)
    p = tmp_path / "in.bin"
    o = dict(ep=np.zeros(0), num_ev_map=np.zeros((w.pano_h, w.pano_w), np.int32),
             ne=dict(P=0, A11=np.zeros((3 * w.K, 3 * w.K)), b1=np.zeros(3 * w.K), A22=np.zeros((0, 2, 2)), b2=np.zeros(0), active=np.zeros(0, np.uint32)))
    _write_case(p, wi, o)           # (only the inputs are read)
    max_iter = 8
    out_dir = tmp_path / "results"
    r = subprocess.run([exe, str(p), devices, str(max_iter), "1" if use_irls else "0", "1" if use_cg else "0", str(out_dir)], capture_output=True, text=True, timeout=300)
    print(r.stdout[-3000:], r.stderr[-2000:])
    assert r.returncode == 0, r.stdout + r.stderr
    lines = r.stdout.strip().splitlines()
    lm = [l.split() for l in lines if l.startswith("LM ")]
    end = [l.split() for l in lines if l.startswith("END ")][0]
    knots = np.array([[float(v) for v in l.split()[1:]] for l in lines if l.startswith("KNOT ")])
    ba = BASettings(use_IRLS=use_irls, cost_type="huber", eta=0.1, alpha=5.0, use_CG=use_cg)
    # (1) the Python resident loop on the device path, one rank: same decisions, same costs
    m = LEGM(w.sensor_w, w.sensor_h, w.lut, w.C_th, w.pano_w, w.pano_h, device=0)
    rp = solve_time_window(m, init, w.events, w.Gx, w.Gy, ba, LMSettings(max_num_iter=max_iter), resident=True)
    m.close()
    assert len(lm) == len(rp.log) and int(end[1]) == rp.iterations and bool(int(end[2])) == rp.converged
    assert [int(l[5]) for l in lm] == [int(e[4]) for e in rp.log], "accept / reject sequence differs from solver.py's"
    ranks = len(devices.split(","))
    tol_py = 1e-2 if use_cg else (1e-12 if ranks == 1 else 1e-9)      # (several ranks: other summation order of the per-rank sums; CG: see below)
    for l, e in zip(lm, rp.log):
        assert float(l[3]) == pytest.approx(e[2], rel=tol_py) and float(l[4]) == pytest.approx(e[3], rel=tol_py)
    # (2) the loop on the CPU oracle
    om = OracleModel(oracle_mod, w, use_cg=use_cg)
    ro = solve_time_window(om, init, w.events, w.Gx, w.Gy, ba, LMSettings(max_num_iter=max_iter))
    assert len(lm) == len(ro.log) and int(end[1]) == ro.iterations and bool(int(end[2])) == ro.converged
    assert [int(l[5]) for l in lm] == [int(e[4]) for e in ro.log], "accept / reject sequence differs from the oracle loop"
    assert any(int(l[5]) for l in lm) and not all(int(l[5]) for l in lm), "the case is meant to contain accepted AND rejected steps"
    tol = 1e-2 if use_cg else 1e-7      # (CG stops at a relative residual of 1e-6, model.cpp:823-824: each LM step's iterate is only that well defined)
    for l, e in zip(lm, ro.log):
        assert float(l[3]) == pytest.approx(e[2], rel=tol) and float(l[4]) == pytest.approx(e[3], rel=tol)
    assert np.abs(knots - ro.traj.knots_xyzw).max() < (1e-3 if use_cg else 1e-7)
    Gx_o, Gy_o = om.downloadMap()
    n = Gx_o.size
    sx = float((Gx_o.ravel() * ((np.arange(n) % 7) + 1)).sum()); sy = float((Gy_o.ravel() * ((np.arange(n) % 5) + 1)).sum())
    mp_ = [l.split() for l in lines if l.startswith("MAP ")][0]
    assert float(mp_[1]) == pytest.approx(sx, rel=5e-2 if use_cg else 1e-7, abs=1e-9) and float(mp_[2]) == pytest.approx(sy, rel=5e-2 if use_cg else 1e-7, abs=1e-9)
    # (3) the reference's run-time records (solver.cpp:105-151, 170-178, 205-223, 271-291)
    fr = out_dir / "final_results"
    it_lines = (fr / "iterations.txt").read_text().splitlines()
    assert it_lines[0] == "window #1" and sum(l.startswith("iter #") for l in it_lines) == len(lm)
    n_acc = sum(int(l[5]) for l in lm)
    form = (fr / "runtime_formEqs.txt").read_text().splitlines()
    assert len(form) == 1 + n_acc - (1 if int(lm[-1][5]) and (int(end[2]) or len(lm) > max_iter) else 0) or len(form) in (n_acc, n_acc + 1)
    assert all(l.startswith("iter #") and "count_formEqs" in l and "sec_average_formEqs" in l for l in form)
    assert len((fr / "runtime_solveEqs.txt").read_text().splitlines()) == len(lm)
    obj = (fr / "runtime_objFuncs.txt").read_text().splitlines()
    assert len(obj) == len(lm) and all(" Np = " in l for l in obj)
    if use_cg:
        assert len((fr / "CG_iterations.txt").read_text().splitlines()) == len(lm)
