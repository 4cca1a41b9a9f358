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
