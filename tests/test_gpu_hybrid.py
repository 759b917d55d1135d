"""SURVEY 8(f) N4, the hybrid CPU-cache / GPU-cache tier: UnifiedCache::HybridInit + PreSCCacheController::HybridInsert
(SS/cache/cache.cu:614-670,138-153; HybridInitPair SS/cache/cache_impl.cuh:113-123) and the lookup kernel feat_cache_lookup
(cache_impl.cuh:202-235) as one more source class of gather_kernel -- against the oracle's restatement of the same lines:
per-GPU hotness order, the id -> slot map, both caches' rows, hit classes and byte-identical gathered rows."""
import numpy as np
import pytest
import torch

from oracle import ffi
from tests.gpu_harness import CpuSide, GpuSide
from tests.helpers import Workload, compare_batches

pytestmark = pytest.mark.gpu

FILL = np.float32(-777.25)          # what a row nobody wrote still holds


def presample(gpu, cpu, wl, batch, P):
    steps = min((wl.sets[(p, 0)][0].size - 1) // batch for p in range(P))
    assert steps >= 1
    for p in range(P):
        for it in range(steps):
            compare_batches(gpu.run(p, it, 0, is_presc=True), cpu.run(p, it, 0, is_presc=True), f"presc gpu {p} it {it}: ")


def last_frontier_topo_hits(gpu, p, g, fanout):
    """tmp_part_ind over the frontier of the last sampler pass (the topology hit mask, cache_impl.cuh:126-142)."""
    H = len(fanout)
    n_f = int(g["edge_counter"][9 + H - 1] - g["edge_counter"][9 + H - 2]) if H > 1 else int(g["node_counter"][9])
    return gpu.pools[p].buffer("tmp_part_ind")[:n_f].cpu().numpy()


def oracle_hybrid(cpu, wl, cpu_cap, gpu_cap):
    """One OracleCache per GPU, each from that GPU's own counters (cache.cu:626-643)."""
    cpu.Kg = 1
    cpu.caches = []
    for p in range(wl.P):
        c = ffi.OracleCache(wl.N, wl.D, 1, p)
        c.hybrid_init(cpu.node_access[p], wl.features, cpu_cap, gpu_cap)
        cpu.caches.append(c)
    return cpu.caches


@pytest.mark.parametrize("dim,P,cpu_cap,gpu_cap", [
    (128, 1, 300, 200),
    (100, 2, 150, 90),          # two GPUs, each with the order of its own counters
    (256, 1, 64, 400),
    (128, 1, 0, 250),           # no CPU cache: every hit is a GPU-cache row
    (100, 1, 250, 0),           # no GPU cache: every hit is a CPU-cache row
    (7, 1, 40, 40),             # rows that are no multiple of 16 bytes
    (128, 1, 5000, 5000),       # capacities beyond N (2048): every vertex is cached, ranks past N are skipped
])
def test_hybrid_maps_caches_and_served_rows(hip, buckets, col_slots, dim, P, cpu_cap, gpu_cap):
    wl = Workload(scale=11, edge_factor=8, dim=dim, partition_count=P, n_seeds=1200)
    fanout, batch = [5, 4], 64
    gpu, cpu = GpuSide(wl, batch, fanout), CpuSide(wl, batch, fanout)
    presample(gpu, cpu, wl, batch, P)
    gpu.cache.hybrid_init(gpu.feature, gpu.graph, cpu_cap, gpu_cap)
    assert all(gpu.graph.column_slots(p) == col_slots for p in range(P))
    caches = oracle_hybrid(cpu, wl, cpu_cap, gpu_cap)
    in_cpu = in_gpu = missed = 0
    n_gpu, n_cpu = min(gpu_cap, wl.N), max(min(cpu_cap, wl.N - gpu_cap), 0)       # ranks past N are skipped
    for p, oc in enumerate(caches):
        # the order, the map (HybridInitPair) and the untouched topology maps
        assert np.array_equal(gpu.cache.array("QF", p).cpu().numpy(), oc.arr("QF", np.int32))
        assert np.array_equal(gpu.cache.array("AF", p).cpu().numpy().view(np.uint64), oc.arr("AF", np.uint64))
        node_map = gpu.cache.array("node_map", p).cpu().numpy()
        assert np.array_equal(node_map, oc.arr("node_map", np.int32))
        assert np.all(gpu.cache.array("edge_index_map", p).cpu().numpy() == -2)
        assert np.all(gpu.cache.array("edge_offset_map", p).cpu().numpy() == -2)
        assert np.array_equal(gpu.cache.array("node_access_time", p).cpu().numpy().view(np.uint64), cpu.node_access[p])   # the counters survive the sort
        qf = oc.arr("QF", np.int32)
        assert np.array_equal(node_map[qf[:n_gpu]], cpu_cap + np.arange(n_gpu))                 # cache_impl.cuh:115-117
        assert np.array_equal(node_map[qf[n_gpu:n_gpu + n_cpu]], np.arange(n_cpu))              # :118-121
        assert np.all(node_map[qf[n_gpu + n_cpu:]] == -2)
        # both caches hold the rows of their ranks, byte for byte
        cpu_rows, gpu_rows = gpu.cache.hybrid_caches(p)
        assert np.array_equal(gpu_rows[:n_gpu].cpu().numpy().view(np.uint32), wl.features[qf[:n_gpu]].view(np.uint32))
        assert np.array_equal(cpu_rows[:n_cpu].cpu().numpy().view(np.uint32), wl.features[qf[n_gpu:n_gpu + n_cpu]].view(np.uint32))
        for mode in (0, 1, 2):
            g, c = gpu.run(p, 0, mode), cpu.run(p, 0, mode)
            compare_batches(g, c, f"serve gpu {p} mode {mode}: ")
            csb = g["cache_search_buffer"]
            assert np.array_equal(csb, c["cache_search_buffer"])                                # hit mask + slots of the last op
            assert np.all(last_frontier_topo_hits(gpu, p, g, fanout) == -2)                     # no cached topology
            # hit classes over every node of the batch (the buffer above holds the last hop's only)
            slots = node_map[g["sampled_ids"]]
            in_cpu += int(((slots >= 0) & (slots < cpu_cap)).sum())
            in_gpu += int((slots >= cpu_cap).sum())
            missed += int((slots < 0).sum())
            # a row of either tier is the vertex's own row (the tiers only decide where it is read from)
            assert np.array_equal(g["float_features"].view(np.uint32), wl.features[g["sampled_ids"]].view(np.uint32))
    assert (in_cpu > 0) == (n_cpu > 0) and (in_gpu > 0) == (n_gpu > 0)
    assert (missed > 0) == (cpu_cap + gpu_cap < wl.N)
    gpu.close(); cpu.close()


@pytest.mark.parametrize("dim", [100, 128, 256])
def test_hybrid_lookup_to_the_letter_leaves_miss_rows_unwritten(hip, col_slots, dim):
    """miss_from_table = False is feat_cache_lookup as written (cache_impl.cuh:224-231): a hit row is copied, a miss row is
    nobody's (the unreleased SSD reader's, operator_impl.cu:522-539) -- the buffer keeps what it held."""
    wl = Workload(scale=11, edge_factor=8, dim=dim, n_seeds=1200)
    fanout, batch, cpu_cap, gpu_cap = [5, 4], 64, 120, 180
    gpu, cpu = GpuSide(wl, batch, fanout), CpuSide(wl, batch, fanout)
    presample(gpu, cpu, wl, batch, 1)
    gpu.cache.hybrid_init(gpu.feature, gpu.graph, cpu_cap, gpu_cap, miss_from_table=False)
    oc = oracle_hybrid(cpu, wl, cpu_cap, gpu_cap)[0]
    node_map = oc.arr("node_map", np.int32)
    for mode in (0, 1):
        gpu.pools[0].buffer("float_features").fill_(float(FILL))
        op = cpu.pools[0].p.contents
        np.ctypeslib.as_array(op.float_features, shape=(int(op.feature_rows) * dim,))[:] = FILL
        g = gpu.run(0, 0, mode)
        ids, labels = wl.sets[(0, mode)]
        cpu.pools[0].run_batch(cpu.graph, oc, None, ids, labels, batch, 0, mode, False)       # no table behind the caches
        c = cpu.pools[0].read_batch()
        compare_batches(g, c, f"mode {mode}: ")
        miss = node_map[g["sampled_ids"]] < 0
        assert miss.any() and (~miss).any()
        assert np.all(g["float_features"][miss] == FILL)
        assert np.array_equal(g["float_features"][~miss].view(np.uint32), wl.features[g["sampled_ids"][~miss]].view(np.uint32))
    gpu.close(); cpu.close()


def test_hybrid_after_a_clique_fill_and_back(hip, col_slots):
    """One cache object, one graph: FillUp (cached topology, striped map) -> HybridInit (row headers back on the full CSR, other map,
    column slots rebuilt) -> FillUp again; every stage serves what the oracle serves."""
    wl = Workload(scale=11, edge_factor=8, dim=32, n_seeds=1200)
    fanout, batch = [5, 4], 64
    gpu, cpu = GpuSide(wl, batch, fanout), CpuSide(wl, batch, fanout)
    presample(gpu, cpu, wl, batch, 1)

    def clique_stage(tag):
        gpu.cache.candidate_selection(0, gpu.graph)
        gpu.cache.set_capacity(220, 130)
        gpu.cache.fill_up(gpu.feature, gpu.graph)
        cpu.build_cache(0, capacity=(220, 130))
        g, c = gpu.run(0, 1, 0), cpu.run(0, 1, 0)
        compare_batches(g, c, tag)
        assert np.array_equal(g["cache_search_buffer"], c["cache_search_buffer"])
        assert (last_frontier_topo_hits(gpu, 0, g, fanout) >= 0).any()                          # cached topology in use

    clique_stage("clique fill: ")
    gpu.cache.hybrid_init(gpu.feature, gpu.graph, 100, 150)
    cpu.graph = ffi.OracleGraph(1, wl.indptr, wl.col)           # no cached CSR attached any more
    oracle_hybrid(cpu, wl, 100, 150)
    g, c = gpu.run(0, 1, 0), cpu.run(0, 1, 0)
    compare_batches(g, c, "hybrid after clique fill: ")
    assert np.array_equal(g["cache_search_buffer"], c["cache_search_buffer"])
    clique_stage("clique fill after hybrid: ")
    gpu.close(); cpu.close()
