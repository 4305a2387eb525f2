import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# muse_run_device hands runs with more than one theta component to the host loop (the faster one there); the tests compare the
# loop KERNEL with the host loop for every ntheta, so they ask for it whatever ntheta (read once by the library)
os.environ.setdefault("MUSE_DEBUG_LOOP_ANY_NTHETA", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def M():
    import museinference_jl_amd as M
    return M


@pytest.fixture(scope="session")
def O():
    """The CPU oracle (test infrastructure)."""
    from oracle import oracle
    oracle.build()
    return oracle


def _hip_device_count():
    """Devices the HIP runtime itself counts (torch.cuda.is_available() was seen to answer False in a process whose first
    HIP call came from libmuse_hip.so -- a test module that does not use this fixture running first)."""
    import ctypes
    for name in ("libamdhip64.so", "/opt/rocm/lib/libamdhip64.so"):
        try:
            n = ctypes.c_int(0)
            if ctypes.CDLL(name).hipGetDeviceCount(ctypes.byref(n)) == 0:
                return n.value
        except OSError:
            continue
    return 0


@pytest.fixture(scope="session")
def gpu(M):
    import torch
    if not torch.cuda.is_available() and _hip_device_count() < 1:
        pytest.skip("no GPU")
    M.load_library()
    return True
