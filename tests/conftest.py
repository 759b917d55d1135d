import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _ensure_built():
    """The native artefacts are built in-tree and kept out of git.  A checkout without them (and with
    hipcc at hand) is built once before the tests run -- the same call the driver's build step makes.
    This is not a fallback of the product: legion_amd itself refuses to work without its library."""
    import glob
    import shutil
    need = [os.path.join(ROOT, "legion_amd", "liblegion_hip.so"),
            os.path.join(ROOT, "legion_amd", "bin", "sampling_server")]
    missing = [p for p in need if not os.path.exists(p)]
    if not glob.glob(os.path.join(ROOT, "legion_amd", "trainer", "ipc_service*.so")):
        missing.append("ipc_service extension")
    if missing and (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        import __graft_entry__
        __graft_entry__.build()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    _ensure_built()


@pytest.fixture(scope="session")
def oracle():
    from oracle import ffi
    return ffi.load()


@pytest.fixture(scope="session")
def hip():
    """The product library on a GPU box; GPU tests fail loudly (never skip) if it is not loadable."""
    import torch
    from legion_amd import lib
    L = lib.load()
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    assert L.legion_device_count() >= 1
    return L
