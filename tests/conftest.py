import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _ensure_built():
    import __graft_entry__
    __graft_entry__.ensure_built()     # a checkout without the in-tree native artefacts is built once


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    _ensure_built()


def pytest_collection_modifyitems(config, items):
    """No GPU test may hang the run: a server or trainer process that dies leaves its peer blocked on a semaphore.
    pytest-timeout (installed in this image) turns that into a failure with a traceback."""
    if not config.pluginmanager.hasplugin("timeout"):
        return
    for item in items:
        if item.get_closest_marker("gpu") is not None and item.get_closest_marker("timeout") is None:
            item.add_marker(pytest.mark.timeout(600))


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


@pytest.fixture(params=["8", "16"], ids=["8-buckets", "16-buckets"])
def buckets(request, monkeypatch):
    """Runs a GPU test with the small class's 8 and with its 16 hash buckets per lane (LegionTuning.lds_small_buckets; pools of
    larger hops have 64 / 256 either way): how many (lane, bucket) workgroups de-duplicate a hop must not change a batch.
    (Rounds 1-4 ran these tests over three forms of the first-touch state -- a uint32[N] array, an open-addressing table, the LDS
    form; since round 5 the LDS form is the only one.)"""
    monkeypatch.setenv("LEGION_LDS_SMALL_BUCKETS", request.param)
    return int(request.param)


@pytest.fixture(params=["0", "1"], ids=["no-column-slots", "column-slots"])
def col_slots(request, monkeypatch):
    """Runs a GPU test without and with the {neighbour id, feature-cache slot} copy of the column array (LegionTuning.col_slots):
    the gather takes a row's cache slot from node_map[id], or from what the sampler carried along.  Same results."""
    monkeypatch.setenv("LEGION_COL_SLOTS", request.param)
    return request.param == "1"
