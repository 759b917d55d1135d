"""The other BASELINE.json configurations as parity / property cases (not bench lines):
products-shaped [25,10] B=1024 all-resident, a 3-hop [15,10,5] run with a hotness cache whose misses
go to mapped pinned host memory (spill-over tier), and full-size size-independent properties."""
import numpy as np
import pytest
import torch

from legion_amd import engine, synth
from tests.gpu_harness import CpuSide, GpuSide
from tests.helpers import Workload, check_invariants, compare_batches

pytestmark = pytest.mark.gpu


def test_products_shaped_all_resident(hip):
    """configs[1]: D = 100 (400-byte rows), batch 1024, [25,10], everything in HBM, no cache-miss path
    (capacity = N: every row is a hit).  Oracle parity on a scaled-down graph of the same shape."""
    wl = Workload(scale=14, edge_factor=50, dim=100, n_seeds=4000)       # products: E/N ~ 50, D = 100
    fanout, batch = [25, 10], 1024
    gpu, cpu = GpuSide(wl, batch, fanout), CpuSide(wl, batch, fanout)
    for it in range(2):
        gpu.run(0, it, 0, is_presc=True); cpu.run(0, it, 0, is_presc=True)
    gpu.cache.candidate_selection(0, gpu.graph)
    gpu.cache.set_capacity(wl.N, 10)                                   # cost-model bypass: whole table cached
    gpu.cache.fill_up(gpu.feature, gpu.graph)
    cpu.build_cache(0, capacity=(wl.N, 10))
    for it in range(2):
        g, c = gpu.run(0, it, 0), cpu.run(0, it, 0)
        compare_batches(g, c, f"products batch {it}: ")
        assert np.array_equal(g["cache_search_buffer"], c["cache_search_buffer"])
        assert (g["cache_search_buffer"] >= 0).all()                   # no miss path
        check_invariants(wl, g, fanout)
    gpu.close(); cpu.close()


def test_three_hops_with_pinned_host_spillover(hip):
    """configs[2]: 3-hop [15,10,5], hotness-ranked cache in HBM, full CSR and full feature table in
    mapped pinned host memory: topology misses and feature misses are read in place over PCIe."""
    wl = Workload(scale=13, edge_factor=12, dim=128, n_seeds=3000)
    fanout, batch = [15, 10, 5], 256
    dev = torch.device("cuda:0")
    pinned = [engine.PinnedArray(wl.indptr), engine.PinnedArray(wl.col), engine.PinnedArray(wl.features)]
    graph = engine.GraphStorage(1, pinned[0].tensor(dev), pinned[1].tensor(dev))
    feature = engine.FeatureStorage(1, pinned[2].tensor(dev))
    ids, labels = wl.sets[(0, 0)]
    feature.set_ids(0, 0, ids, labels)
    cache = engine.UnifiedCache(800_000, wl.D, 4, 1, wl.N)
    cache.init_controller(0)
    pool = engine.MemoryPool(0, wl.N, batch, fanout, wl.D)
    cpu = CpuSide(wl, batch, fanout)
    for it in range(4):
        engine.enqueue_batch(None, graph, feature, cache, pool, batch, it, 0, 0, True, fanout)
        cpu.run(0, it, 0, is_presc=True)
    torch.cuda.synchronize()
    cache.candidate_selection(0, graph)
    cache.cost_model(feature, graph, (0, 0), 4)
    cache.fill_up(feature, graph)
    oc = cpu.build_cache(0, cache_memory=800_000, train_step=4)[0]
    assert (cache.node_capacity(0), cache.edge_capacity(0)) == (oc.node_capacity, oc.edge_capacity)
    pool.alloc_features(pool.num_ids)
    hits = misses = 0
    for it in range(3):
        engine.enqueue_batch(None, graph, feature, cache, pool, batch, it, 0, 0, False, fanout)
        torch.cuda.synchronize()
        g, c = engine.read_batch(pool), cpu.run(0, it, 0)
        compare_batches(g, c, f"3-hop spill batch {it}: ")
        ci = pool.buffer("cache_search_buffer")[:int(g["node_counter"][1])].cpu().numpy()
        hits += int((ci >= 0).sum()); misses += int((ci < 0).sum())
    assert hits > 0 and misses > 0                                      # both tiers were exercised
    pool.close(); cache.close(); feature.close(); graph.close()
    for p in pinned:
        p.close()
    cpu.close()


@pytest.mark.parametrize("scale,batch,group", [(22, 8000, 4), (26, 1024, 16)], ids=["rmat22-b8000", "rmat26-b1024-full-size"])
def test_full_size_properties(hip, scale, batch, group):
    """Size-independent properties on graphs far beyond what the oracle replays in seconds -- RMAT-22
    (67 M edges) at B = 8000 and BASELINE.json's full-size workload, RMAT-26 (1.07 G edges, 34 GB of
    features) at B = 1024, lane groups with graph replay: gathered rows recomputed from the generator bit
    for bit, unique ids, localisation, per-hop edge counts = sum of min(fan-out, degree)."""
    D, fanout = 128, [25, 10]
    N = 1 << scale
    dev = torch.device("cuda:0")
    indptr, col = synth.rmat_csr_device(scale, 16, 20231, dev)
    feats = synth.features_device(N, D, 7, dev)
    seeds = synth.seed_ids(N, 200_000, 11)
    graph, feature = engine.GraphStorage(1, indptr, col), engine.FeatureStorage(1, feats)
    feature.set_ids(0, 0, seeds, None)
    cache = engine.UnifiedCache(64 << 20, D, 8, 1, N)
    cache.init_controller(0)
    pool = engine.MemoryPool(0, N, batch, fanout, D)
    for it in range(8):
        engine.enqueue_batch(None, graph, feature, cache, pool, batch, it, 0, 0, True, fanout)
    torch.cuda.synchronize()
    cache.candidate_selection(0, graph)
    cache.cost_model(feature, graph, (0, 0), 8)
    cache.fill_up(feature, graph)
    rows = int(cache.max_id_num(0) * 1.2)
    pipe = engine.Pipeline(graph, feature, cache, 0, batch, fanout, group, rows, True, 2)
    deg = (indptr[1:] - indptr[:-1])
    for c0 in (0, group, 2 * group):
        slot = pipe.submit(c0)
        pipe.wait(slot)
        for lane in range(0, group, max(1, group // 4)):
            pl = pipe.pools[slot][lane]
            nc = pl.buffer("node_counter").cpu().numpy(); ec = pl.buffer("edge_counter").cpu().numpy()
            n, e = int(nc[11]), int(ec[11])
            assert n <= rows
            ids = pl.buffer("sampled_ids")[:n]
            assert int(torch.unique(ids).numel()) == n
            assert synth.feature_check_device(pl.buffer("float_features")[:n].contiguous(), ids.contiguous(), D, 7) == 0
            src_g, dst_g = pl.buffer("agg_src_ids")[:e].long(), pl.buffer("agg_dst_ids")[:e].long()
            assert bool((ids.long()[pl.buffer("agg_src_off")[:e].long()] == src_g).all())
            assert bool((ids.long()[pl.buffer("agg_dst_off")[:e].long()] == dst_g).all())
            b = int(nc[9])
            assert np.array_equal(ids[:b].cpu().numpy(), seeds[(c0 + lane) * batch:(c0 + lane + 1) * batch])
            e1 = int(ec[10])
            assert e1 == int(torch.clamp(deg[ids[:b].long()], max=fanout[0]).sum())
            assert e - e1 == int(torch.clamp(deg[src_g[:e1]], max=fanout[1]).sum())
            # every sampled neighbour is a real neighbour (spot check on the device)
            k = torch.randint(0, e, (2000,), device=dev)
            for kk in k[:50].tolist():
                row = col[int(indptr[dst_g[kk]]):int(indptr[dst_g[kk] + 1])]
                assert bool((row == src_g[kk]).any())
            hit = pl.buffer("cache_search_buffer")[:int(nc[1])]
            assert int((hit >= 0).sum()) > 0
    pipe.close(); pool.close(); cache.close(); feature.close(); graph.close()


def test_refill_and_second_cache_with_column_slots(hip, monkeypatch):
    """Column slots ({neighbour id, feature-cache slot} pairs of the column array) belong to ONE fill of ONE cache: after a
    re-fill with another capacity, or with a second cache object over the same graph, the sampler must not carry slots of the
    old node_map into the gather (stale pairs gave wrong rows, or out-of-range reads when the capacity shrank).
    fill -> serve -> set_capacity -> fill -> serve, then a second cache serving from the same graph: all bit-exact."""
    monkeypatch.setenv("LEGION_COL_SLOTS", "1")
    wl = Workload(scale=11, edge_factor=8, dim=24, n_seeds=1200)
    fanout, batch = [6, 4], 64
    gpu, cpu = GpuSide(wl, batch, fanout), CpuSide(wl, batch, fanout)
    for it in range(4):
        gpu.run(0, it, 0, is_presc=True); cpu.run(0, it, 0, is_presc=True)
    gpu.cache.candidate_selection(0, gpu.graph)
    for capacity in ((400, 50), (90, 20), (700, 10)):            # shrink, then grow
        gpu.cache.set_capacity(*capacity)
        gpu.cache.fill_up(gpu.feature, gpu.graph)
        assert gpu.graph.column_slots(0)
        cpu.build_cache(0, capacity=capacity)
        hits = 0
        for it in range(3):
            g, c = gpu.run(0, it, 0), cpu.run(0, it, 0)
            compare_batches(g, c, f"capacity {capacity} batch {it}: ")
            assert np.array_equal(g["cache_search_buffer"], c["cache_search_buffer"])
            hits += int((g["cache_search_buffer"] >= 0).sum())
        assert hits > 0
    # a second cache over the same graph, filled with yet another capacity: the graph's pairs now belong to IT ...
    cache2 = engine.UnifiedCache(0, wl.D, 1, 1, wl.N)
    cache2.init_controller(0)
    pool2 = engine.MemoryPool(0, wl.N, batch, fanout, wl.D)
    pool2.alloc_features(pool2.num_ids)
    for it in range(4):
        engine.enqueue_batch(None, gpu.graph, gpu.feature, cache2, pool2, batch, it, 0, 0, True, fanout)
    torch.cuda.synchronize()
    cache2.candidate_selection(0, gpu.graph)
    cache2.set_capacity(33, 10)                  # (a graph holds ONE topology cache: the same vertices as the first cache's last fill)
    cache2.fill_up(gpu.feature, gpu.graph)
    cpu2 = CpuSide(wl, batch, fanout)
    for it in range(4):
        cpu2.run(0, it, 0, is_presc=True)
    cpu2.build_cache(0, capacity=(33, 10))
    engine.enqueue_batch(None, gpu.graph, gpu.feature, cache2, pool2, batch, 1, 0, 0, False, fanout)
    torch.cuda.synchronize()
    compare_batches(engine.read_batch(pool2), cpu2.run(0, 1, 0), "second cache: ")
    # ... and the FIRST cache (capacity 700), whose pairs are gone, still serves correctly: it looks its slots up
    cpu.build_cache(0, capacity=(700, 10))       # (the oracle graph's attached cache is the latest one built: rebuild)
    g, c = gpu.run(0, 2, 0), cpu.run(0, 2, 0)
    compare_batches(g, c, "first cache after the second fill: ")
    assert np.array_equal(g["cache_search_buffer"], c["cache_search_buffer"])
    pool2.close(); cache2.close(); cpu2.close()
    gpu.close(); cpu.close()


@pytest.mark.parametrize("D", [7, 100, 602])
def test_feature_cache_row_widths_in_a_striped_clique(hip, col_slots, D):
    """Rows of the HBM-resident feature cache (own stripe, the other member's stripe, a hot-row replica) at widths that are not
    whole 128-byte lines: D = 7 (28-byte rows, scalar path), 100 (products: 400 bytes), 602 (2408 bytes, unaligned 16-byte chunks
    + tail); byte-identical with the oracle (SS/cache/cache_impl.cuh:259-268)."""
    pitch = "dense"
    wl = Workload(scale=11, edge_factor=8, dim=D, partition_count=2, n_seeds=1200)
    fanout, batch = [5, 4], 64
    gpu, cpu = GpuSide(wl, batch, fanout), CpuSide(wl, batch, fanout)
    for p in range(2):
        for it in range(3):
            gpu.run(p, it, 0, is_presc=True); cpu.run(p, it, 0, is_presc=True)
    gpu.cache.candidate_selection(1, gpu.graph)               # one clique of two: own stripe + the other member's
    gpu.cache.set_replica_memory(40 * D * 4)                  # + a 40-row (or fewer, padded) replica
    gpu.cache.set_capacity(300, 40)
    gpu.cache.fill_up(gpu.feature, gpu.graph)
    cpu.build_cache(1, capacity=(300, 40))
    hits = 0
    for p in range(2):
        for it in range(2):
            g, c = gpu.run(p, it, 0), cpu.run(p, it, 0)
            compare_batches(g, c, f"D={D} {pitch} gpu {p} batch {it}: ")
            assert np.array_equal(g["cache_search_buffer"], c["cache_search_buffer"])
            hits += int((g["cache_search_buffer"] >= 0).sum())
    assert hits > 0
    gpu.close(); cpu.close()
