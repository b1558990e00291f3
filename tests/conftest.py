"""pytest configuration: markers, import paths and shared fixtures.

`-m "not gpu"` runs on CPU only (oracle vs golden vectors, host plan/designer logic through
the C ABI, symbol export checks, gloo multi-process path).  `-m gpu` tests are the parity tests
proper: they call the HIP path through the C-ABI shared library and compare with the oracle.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU test")


@pytest.fixture(scope="session")
def golden_streams():
    return np.load(os.path.join(GOLDEN, "streams.npz"))


@pytest.fixture(scope="session")
def golden_tables():
    return np.load(os.path.join(GOLDEN, "tables.npz"))


@pytest.fixture(scope="session")
def refwrap():
    """The REAL reference (oracle/_ref); skips when it was not built/shipped."""
    import refwrap as R
    if not R.available():
        pytest.skip("oracle/_ref not built (needs /root/reference; see oracle/Makefile)")
    return R


def rms(a):
    a = np.asarray(a, dtype=np.float64)
    return float(np.sqrt(np.mean(a * a))) if a.size else 0.0


def peak(a):
    a = np.asarray(a)
    return float(np.max(np.abs(a))) if a.size else 0.0
