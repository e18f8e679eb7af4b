import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_mod():
    """The CPU oracle (test infrastructure).  Built on demand with gcc."""
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def hip_lib():
    """libemba_hip.so, built in-tree with hipcc if missing (cross-compiles without a GPU)."""
    from emba_amd import build, _lib
    build.build_hip()
    return _lib.load()


def pytest_terminal_summary(terminalreporter):
    """Worst ELEMENT-WISE errors per quantity over the session (tests/helpers.py: assert_close_elementwise), for the log and BASELINE.md."""
    try:
        import helpers
    except Exception:   # noqa: BLE001
        return
    if not helpers.WORST:
        return
    terminalreporter.write_line("worst element-wise error per quantity, |a-b| / (|b| + 1e-7 max|b|)  [bound 1e-5]:")
    for k in sorted(helpers.WORST):
        terminalreporter.write_line(f"    {k:28s} {helpers.WORST[k]:.3e}")
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        import json
        with open(os.path.join(out, "parity_elementwise.json"), "w") as f:
            json.dump(helpers.WORST, f, indent=1, sort_keys=True)
