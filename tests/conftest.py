import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

# the reference's constants (main_rt.py:449-457)
D_PLANE = float(np.float64(0.12156646438729327) + np.float64(0.08843353561270673))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # GPU sessions mix the NumPy API and the torch-tensor API in one process: load PyTorch's HIP runtime first (the
    # library itself no longer imports torch: _lib.lib()).
    if "gpu" in (config.getoption("-m") or "") and "not gpu" not in (config.getoption("-m") or ""):
        import torch  # noqa: F401


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def rtus():
    import rtus as _rtus
    if not os.path.exists(_rtus.LIB_PATH):          # fresh checkout: compile librtus.so (hipcc cross-compiles gfx950 without a GPU)
        _rtus.build()
    return _rtus


def nan_equal_mask(a, b):
    return np.array_equal(np.isnan(a), np.isnan(b))


def max_abs(a, b):
    m = ~(np.isnan(a) | np.isnan(b))
    return float(np.max(np.abs(a[m] - b[m]))) if m.any() else 0.0
