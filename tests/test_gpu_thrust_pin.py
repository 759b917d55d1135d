"""Pins the third-party semantics the reference relies on -- on the DEVICE paths it actually ran.

oracle/thrust_device_pin.hip executes the reference's own rocThrust device expressions on the MI355X
(sort_by_key(thrust::device, ..., greater<u64>) SS/cache/cache.cu:415,435; inclusive_scan :471-472,500;
minstd_rand + uniform_int_distribution inside a kernel SS/engine/operator_impl.cu:235-238).  Their results are
compared with the product's rocPRIM / table-driven path and with the oracle at N = 2^24 with massive ties
(most vertices tie at hotness 0, as after a PreSC epoch).  This does not make parity "pinned to the
reference" (the reference has no vectors); it closes the last pin that can be closed here."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch

from legion_amd import engine
from oracle import ffi

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PIN = os.path.join(ROOT, "oracle", "_build", "thrust_device_pin")


def test_thrust_device_paths_match_product_and_oracle(hip, oracle, tmp_path):
    if not os.path.exists(PIN):
        ffi.build()
    n, m = 1 << 24, 1 << 20
    rng = np.random.RandomState(5)
    keys = np.floor(rng.pareto(1.1, n) * 0.35).astype(np.uint64)          # ~75 % zeros, long tail, many small ties
    keys[rng.randint(0, n, 1000)] = rng.randint(1 << 20, 1 << 40, 1000).astype(np.uint64)
    assert (keys == 0).mean() > 0.6 and np.unique(keys).size > 200
    idx = np.concatenate([rng.randint(0, 6_000_000, m - 4096), rng.randint(0, 2**31 - 2, 4096)]).astype(np.int32)
    deg = np.concatenate([rng.randint(1, 1000, m // 2), rng.randint(1, 1 << 24, m - m // 2)]).astype(np.int32)
    pairs = np.stack([idx, deg], axis=1).astype(np.int32)
    f = {k: str(tmp_path / k) for k in ("keys", "order", "sorted", "scan", "pairs", "draws")}
    keys.tofile(f["keys"]); pairs.tofile(f["pairs"])
    out = subprocess.run([PIN, f["keys"], str(n), f["order"], f["sorted"], f["scan"], f["pairs"], str(m), f["draws"]],
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert out.returncode == 0, out.stdout
    t_order = np.fromfile(f["order"], dtype=np.int32)
    t_sorted = np.fromfile(f["sorted"], dtype=np.uint64)
    t_scan = np.fromfile(f["scan"], dtype=np.uint64)
    t_draws = np.fromfile(f["draws"], dtype=np.int32)

    # ---- the hotness order: stable descending (ties keep ascending vertex id) -- numpy statement of the oracle's rule
    want_order = np.argsort(np.uint64(0xFFFFFFFFFFFFFFFF) - keys, kind="stable").astype(np.int32)
    assert np.array_equal(t_order, want_order), "rocThrust device sort_by_key(greater) is not stable-descending"
    assert np.array_equal(t_sorted, keys[want_order])
    # ---- the oracle (C)
    oc = ffi.OracleCache(n, 1, Kg=1)
    oc.candidate_selection([keys], [keys])
    assert np.array_equal(oc.arr("QF", np.int32), t_order) and np.array_equal(oc.arr("AF", np.uint64), t_sorted)
    oc.close()
    # ---- the product (rocPRIM radix sort inside CandidateSelection, kernels_cache.hip)
    dev = torch.device("cuda:0")
    indptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    col = torch.zeros(16, dtype=torch.int32, device=dev)
    graph = engine.GraphStorage(1, indptr, col)
    cache = engine.UnifiedCache(1 << 20, 1, 1, 1, n)
    cache.init_controller(0)
    kt = torch.from_numpy(keys.view(np.int64)).to(dev)
    cache.array("node_access_time", 0).copy_(kt)
    cache.array("edge_access_time", 0).copy_(kt)
    cache.candidate_selection(0, graph)
    torch.cuda.synchronize()
    assert np.array_equal(cache.array("QF", 0).cpu().numpy(), t_order)
    assert np.array_equal(cache.array("QT", 0).cpu().numpy(), t_order)
    assert np.array_equal(cache.array("AF", 0).cpu().numpy().view(np.uint64), t_sorted)
    cache.close(); graph.close()

    # ---- inclusive_scan: plain running sums in uint64 (what the cost model's prefix arrays hold, cache.cu:471-472)
    assert np.array_equal(t_scan, np.cumsum(t_sorted, dtype=np.uint64))

    # ---- the draw, executed by rocThrust inside a kernel, against the product's table-driven kernel and the oracle
    it, dt = torch.from_numpy(idx).to(dev), torch.from_numpy(deg).to(dev)
    got = torch.empty(m, dtype=torch.int32, device=dev)
    hip.legion_draw_batch(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), ctypes.c_void_p(it.data_ptr()),
                          ctypes.c_void_p(dt.data_ptr()), ctypes.c_void_p(got.data_ptr()), m)
    torch.cuda.synchronize()
    assert np.array_equal(got.cpu().numpy(), t_draws)
    assert t_draws.min() >= 0 and bool((t_draws < deg).all())
    for j in list(range(0, m, 97)) + list(range(m - 4096, m)):
        assert oracle.lgo_draw(int(idx[j]), int(deg[j])) == int(t_draws[j])
