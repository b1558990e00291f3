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


@pytest.fixture(scope="session")
def hip_hooks():
    """TEST build of the HIP library with the parity-test hooks compiled in (csrc/Makefile `testhooks`:
    tests/_build/libr8bsrc_hip_testhooks.so -- same kernels, r8b_design_set_lp_provider added); the product library has
    neither the symbol nor its code.  Built by __graft_entry__.build(); it travels to the GPU box with the snapshot."""
    import importlib
    path = os.path.join(ROOT, "tests", "_build", "libr8bsrc_hip_testhooks.so")
    if not os.path.exists(path):
        pytest.skip("tests/_build/libr8bsrc_hip_testhooks.so not built (python __graft_entry__.py)")
    r8b = importlib.import_module("r8brain-free-src_amd")
    r8b.load()   # (torch's HIP runtime first, then the product, then the test build: _capi.load)
    return r8b.bind(path, test_hooks=True)


def rms(a):
    a = np.asarray(a, dtype=np.float64)
    return float(np.sqrt(np.mean(a * a))) if a.size else 0.0


def peak(a):
    a = np.asarray(a)
    return float(np.max(np.abs(a))) if a.size else 0.0
