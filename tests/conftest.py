import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
# the product package (mirror of the reference's `onpolicy` import paths) and the oracle
for p in (os.path.join(ROOT, "on-policy_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped (not failed) when no device is present, e.g. `pytest tests/` here."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _gemm_tuning_stays_local(monkeypatch):
    """The train scripts switch PyTorch's GEMM autotuner on (onpolicy/utils/gemm_tuning.py); in the test session
    that would tune every small shape of every later test.  Tests that want it set MAPPO_GEMM_TUNING=1
    themselves; whatever a test enabled is switched off again afterwards."""
    monkeypatch.setenv("MAPPO_GEMM_TUNING", os.environ.get("MAPPO_TEST_GEMM_TUNING", "0"))
    yield
    import torch
    if torch.cuda.is_available() and torch.cuda.tunable.is_enabled():
        torch.cuda.tunable.enable(False)


@pytest.fixture(scope="session")
def gold():
    class G(object):
        def __init__(self):
            self._c = {}

        def npz(self, name):
            if name not in self._c:
                self._c[name] = np.load(os.path.join(GOLD, name + ".npz"))
            return self._c[name]

        def meta(self, name):
            with open(os.path.join(GOLD, name + ".json")) as f:
                return json.load(f)
    return G()
