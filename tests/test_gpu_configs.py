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
