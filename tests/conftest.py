import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "fft-implementation-in-c_amd"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _ensure_product_lib():
    """The C-ABI library is a build product (git-ignored).  Tests never fall back to anything else: if it is not
    there they BUILD it (hipcc cross-compiles gfx950 without a GPU), exactly like __graft_entry__.build()."""
    import subprocess
    pkg = os.path.join(ROOT, "fft-implementation-in-c_amd")
    if not (os.path.exists(os.path.join(pkg, "libfft_mi355x.so")) and os.path.exists(os.path.join(pkg, "libfft_mi355x_exp.so"))):
        subprocess.run(["make", "-s", "-j2", "-C", pkg], check=True)


def pytest_sessionstart(session):
    _ensure_product_lib()
    # the CPU emulation of the kernels (tests/emu): built ONCE here, before any multi-process test has its ranks ask for it
    # (GPU-only sessions do not need it and skip the g++ run)
    if "gpu" not in (session.config.getoption("-m") or "") or "not gpu" in (session.config.getoption("-m") or ""):
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import emu_lib
        emu_lib.lib()


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "reference_vectors.npz"))


@pytest.fixture(scope="session")
def gpu_lib():
    """The product library, initialised on the GPU.  No fallback: a missing
    .so or a missing device is a hard failure of every GPU test."""
    import fftlib
    lib = fftlib.init()
    return lib


@pytest.fixture(autouse=True)
def _default_planner_policy(request):
    """GPU tests steer the planner with fftlib.set_policy(); every test starts from and leaves the default policy."""
    yield
    if request.node.get_closest_marker("gpu") is not None:
        import fftlib
        if fftlib._lib is not None:
            fftlib.set_policy(team=1, min_batch=0, chunk_mb=0)
