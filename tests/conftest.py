import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "recommend-tf2.0_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture
def force():
    """force("dense", "b"): a kernel variant the shape would not select (rec_debug_force: the library reads no environment
    variable); every forced key is cleared when the test ends"""
    from recamd._lib import C
    keys = []

    def _force(key, value):
        keys.append(key)
        C.debug_force(key, value)
    yield _force
    for k in keys:
        C.debug_force(k, None)


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def pytest_sessionfinish(session, exitstatus):
    """parity margins of this session (tests/util.py::MARGINS) -> gpurun_out/parity_margins.json"""
    try:
        from tests.util import MARGINS
    except Exception:  # noqa: BLE001
        return
    if not MARGINS:
        return
    import json
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    worst = sorted(MARGINS.items(), key=lambda kv: -kv[1]["used"])
    with open(os.path.join(out, "parity_margins.json"), "w") as f:
        json.dump({"note": "used = max over the test's checks of err / bound (1.0 = at the tolerance); tests/util.py",
                   "tests": len(MARGINS), "worst": [{"test": k, **v} for k, v in worst[:25]],
                   "all": {k: v for k, v in sorted(MARGINS.items())}}, f, indent=1)
