"""oracle/_ref: the ONE piece of the reference that compiles in this image -- the kernels of GraphStorage::GraphCache,
sampling_server/src/storage/graph_storage_impl.cuh:33-53 (GetNeighborCount, TopoFillUp), built by hipcc from where the file lies
under /root/reference (oracle/Makefile target `ref`, in the build container; the binary travels to the GPU box) behind this repo's
driver, which restates the host wrapper's call sequence (SS/storage/graph_storage.cu:81-104).  What those kernels produce on the GPU is
compared with the oracle's restatement (lgo_fill_up's cached CSR) AND with the product's cached CSR after FillUp: for this row of
SURVEY 8 (a11 / N1: the topology cache's fill) parity is pinned to the reference's code itself, not to a reading of it."""
import os
import struct
import subprocess

import numpy as np
import pytest
import torch

from oracle import ffi
from tests.gpu_harness import CpuSide, GpuSide
from tests.helpers import Workload

pytestmark = pytest.mark.gpu


def run_reference_kernels(tmp_path, QT, indptr, col, Kg, capacity):
    if not os.path.exists(ffi.REF_GRAPH_CACHE):
        why = (f"{ffi.REF_GRAPH_CACHE} is missing: it is built where /root/reference exists (python __graft_entry__.py build, or "
               "make -C oracle ref) and travels to the GPU box with the snapshot")
        assert not os.path.exists("/root/reference"), why          # the reference is here: the binary should have been built
        pytest.skip(why)                                           # a box that never saw the reference: this checker cannot exist there
    fin, fout = str(tmp_path / "ref_in.bin"), str(tmp_path / "ref_out.bin")
    with open(fin, "wb") as f:
        f.write(struct.pack("<4q", QT.size, col.size, Kg, capacity))
        f.write(np.ascontiguousarray(QT, dtype=np.int32).tobytes())
        f.write(np.ascontiguousarray(indptr, dtype=np.int64).tobytes())
        f.write(np.ascontiguousarray(col, dtype=np.int32).tobytes())
    res = subprocess.run([ffi.REF_GRAPH_CACHE, fin, fout], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-2000:]
    raw = open(fout, "rb").read()
    out, off = [], 0
    for _ in range(Kg):
        index = np.frombuffer(raw, dtype=np.int64, count=capacity + 1, offset=off)
        off += (capacity + 1) * 8
        n = int(np.frombuffer(raw, dtype=np.int64, count=1, offset=off)[0])
        off += 8
        dst = np.frombuffer(raw, dtype=np.int32, count=n, offset=off)
        off += n * 4
        assert n == index[-1]
        out.append((index, dst))
    assert off == len(raw)
    return out


@pytest.mark.parametrize("P,mode_bits,capacity", [(1, 0, (300, 200)), (2, 1, (150, 90)), (4, 2, (64, 33)), (8, 3, (40, 20)), (4, 1, (100, 50))])
def test_reference_kernels_oracle_and_product_build_the_same_cached_csr(hip, tmp_path, P, mode_bits, capacity):
    wl = Workload(scale=11, edge_factor=8, dim=8, partition_count=P, n_seeds=1200)
    fanout, batch = [5, 4], 64
    gpu, cpu = GpuSide(wl, batch, fanout), CpuSide(wl, batch, fanout)
    steps = min((wl.sets[(p, 0)][0].size - 1) // batch for p in range(P))
    for p in range(P):
        for it in range(steps):
            gpu.run(p, it, 0, is_presc=True); cpu.run(p, it, 0, is_presc=True)
    gpu.cache.candidate_selection(mode_bits, gpu.graph)
    gpu.cache.set_capacity(*capacity)
    gpu.cache.fill_up(gpu.feature, gpu.graph)
    caches = cpu.build_cache(mode_bits, capacity=capacity)
    Kg, ecap = cpu.Kg, capacity[1]
    for ki, oc in enumerate(caches):
        QT = oc.arr("QT", np.int32)
        assert np.array_equal(gpu.cache.array("QT", ki * Kg).cpu().numpy(), QT)
        ref = run_reference_kernels(tmp_path, QT, wl.indptr, wl.col, Kg, ecap)          # the reference's own kernels, on this GPU
        c = oc.c.contents
        for j in range(Kg):
            r_index, r_dst = ref[j]
            # oracle (lgo_fill_up, graph_storage.cu:81-104 restated) == reference kernels
            o_index = np.ctypeslib.as_array(c.topo_indptr[j], shape=(ecap + 1,))
            assert np.array_equal(o_index, r_index), f"clique {ki} member {j}: oracle index"
            o_dst = np.ctypeslib.as_array(c.topo_col[j], shape=(max(int(o_index[-1]), 1),))[:int(o_index[-1])]
            assert np.array_equal(o_dst, r_dst), f"clique {ki} member {j}: oracle columns"
            # product (storage.hip GraphCacheBuildLocal: topo_neighbor_count + rocPRIM scan + wave-per-vertex topo_fill_up) == reference kernels
            g_index, g_dst = gpu.graph.cached_csr(ki * Kg + j, ecap)
            assert np.array_equal(g_index.cpu().numpy(), r_index), f"clique {ki} member {j}: product index"
            assert np.array_equal(g_dst.cpu().numpy(), r_dst), f"clique {ki} member {j}: product columns"
            assert r_index[-1] > 0
    gpu.close(); cpu.close()


def test_reference_kernels_on_a_skewed_graph_with_empty_rows(hip, tmp_path):
    """Degree-0 vertices, a hub, a capacity that takes every vertex: the reference's kernels against a numpy statement of
    graph_storage_impl.cuh:33-53 and against the oracle."""
    rng = np.random.RandomState(4)
    N = 600
    deg = rng.zipf(1.7, N) % 40
    deg[rng.rand(N) < 0.3] = 0
    deg[17] = 300
    indptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
    col = rng.randint(0, N, size=int(indptr[-1])).astype(np.int32)
    QT = rng.permutation(N).astype(np.int32)
    for Kg, cap in ((1, N), (3, 200), (4, 150), (8, 75)):
        ref = run_reference_kernels(tmp_path, QT, indptr, col, Kg, cap)
        for j in range(Kg):
            ids = QT[np.arange(cap) * Kg + j]
            index = np.concatenate([[0], np.cumsum(deg[ids])]).astype(np.int64)
            dst = np.concatenate([col[indptr[v]:indptr[v + 1]] for v in ids]) if cap else np.zeros(0, np.int32)
            assert np.array_equal(ref[j][0], index) and np.array_equal(ref[j][1], dst.astype(np.int32))
        oc = ffi.OracleCache(N, 0, Kg, 0)
        np.ctypeslib.as_array(oc.c.contents.QT, shape=(N,))[:] = QT
        oc.set_capacity(0, cap)
        oc.fill_up(None, indptr, col)
        c = oc.c.contents
        for j in range(Kg):
            o_index = np.ctypeslib.as_array(c.topo_indptr[j], shape=(cap + 1,))
            assert np.array_equal(o_index, ref[j][0])
            n = int(o_index[-1])
            assert np.array_equal(np.ctypeslib.as_array(c.topo_col[j], shape=(max(n, 1),))[:n], ref[j][1])
        oc.close()
