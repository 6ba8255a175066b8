import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
PKG = os.path.join(REPO, "vit-rpe-rope_amd")
if PKG not in sys.path:
    sys.path.insert(0, PKG)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    class _G:
        def __init__(self):
            self._c = {}

        def __call__(self, name):
            if name not in self._c:
                self._c[name] = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
            return self._c[name]
    return _G()


def rel_err(a, b):
    """max |a-b| / max(|b|) -- the 'rel' of the 1e-4 parity gate."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))
