"""BASELINE.json configs[2..4] at their REAL shapes on one MI355X.

Two kinds of case per config:
  * oracle replay -- the config's batch size, fan-out, feature width and clique layout (8 logical GPUs of one
    device striped as one clique of Kg = 8, topology cache sized by the cost model from the measured PreSC
    transactions) on a graph the C oracle replays in seconds (RMAT-16/17): bit-exact batches, hit masks,
    hotness order, the three id->slot maps, capacities;
  * full size -- the same combination on a graph far beyond the oracle (RMAT-24 / 2^28 vertices / pinned-host
    tables of 2^26 vertices), lane groups + hipGraph replay, checked through size-independent properties
    (rows recomputed from the generator bit for bit, unique ids, localisation, edge counts, real neighbours,
    hits served from peers' stripes and from the topology cache).
Reference shapes: legion_server.py:41-72 (dataset table), SS/cache/cache_impl.cuh:89-109,262-269 (striping),
SS/storage/graph_storage.cu:76-111 (topology cache)."""
import numpy as np
import pytest
import torch

from legion_amd import engine, synth
from tests.gpu_harness import CpuSide, GpuSide
from tests.helpers import Workload, check_invariants, compare_batches

pytestmark = pytest.mark.gpu


def _clique_replay(wl, fanout, batch, mode_bits, cache_memory, serve_batches=1, modes=(0, 1)):
    """PreSC epoch on every logical GPU -> hotness -> order -> cost model fed with the transactions the sampler
    counted -> fills -> serving, GPU vs oracle.  Returns (hits, topo_hits, peer_hits, caps)."""
    P = wl.P
    rows = min(wl.N + 8, int(batch * (1 + sum(np.cumprod(fanout)))))
    gpu, cpu = GpuSide(wl, batch, fanout, cache_memory=cache_memory, feature_rows=rows), CpuSide(wl, batch, fanout, feature_rows=rows)
    steps = min((wl.sets[(p, 0)][0].size - 1) // batch for p in range(P))
    assert steps >= 1
    for p in range(P):
        for it in range(steps):
            g, c = gpu.run(p, it, 0, is_presc=True), cpu.run(p, it, 0, is_presc=True)
            compare_batches(g, c, f"presc gpu {p} it {it}: ")
    tx = 0
    for p in range(P):
        assert np.array_equal(gpu.cache.array("node_access_time", p).cpu().numpy().view(np.uint64), cpu.node_access[p])
        assert np.array_equal(gpu.cache.array("edge_access_time", p).cpu().numpy().view(np.uint64), cpu.edge_access[p])
        assert gpu.cache.max_id_num(p) == cpu.max_ids[p]
        tx += gpu.cache.topo_transactions(p)
    assert tx > 0
    gpu.cache.candidate_selection(mode_bits, gpu.graph)
    gpu.cache.cost_model(gpu.feature, gpu.graph, (tx, 0), steps)
    caches = cpu.build_cache(mode_bits, cache_memory=cache_memory, train_step=steps, counters=(tx, 0))
    Kg = cpu.Kg
    for ki, oc in enumerate(caches):
        for j in range(Kg):
            assert (gpu.cache.node_capacity(ki * Kg + j), gpu.cache.edge_capacity(ki * Kg + j)) == (oc.node_capacity, oc.edge_capacity)
    caps = (caches[0].node_capacity, caches[0].edge_capacity)
    gpu.cache.fill_up(gpu.feature, gpu.graph)
    for ki, oc in enumerate(caches):
        lead = ki * Kg
        assert np.array_equal(gpu.cache.array("QF", lead).cpu().numpy(), oc.arr("QF", np.int32))
        assert np.array_equal(gpu.cache.array("QT", lead).cpu().numpy(), oc.arr("QT", np.int32))
        for j in range(Kg):
            assert np.array_equal(gpu.cache.array("node_map", lead + j).cpu().numpy(), oc.arr("node_map", np.int32))
            assert np.array_equal(gpu.cache.array("edge_index_map", lead + j).cpu().numpy(), oc.arr("edge_index_map", np.int8))
            assert np.array_equal(gpu.cache.array("edge_offset_map", lead + j).cpu().numpy(), oc.arr("edge_offset_map", np.int32))
    hits = topo_hits = peer_hits = 0
    H = len(fanout)
    for p in range(P):
        for mode in modes:
            for it in range(serve_batches if mode == 0 else 1):
                g, c = gpu.run(p, it, mode), cpu.run(p, it, mode)
                compare_batches(g, c, f"serve gpu {p} mode {mode} batch {it}: ")
                assert np.array_equal(g["cache_search_buffer"], c["cache_search_buffer"])      # feature hit mask + slots
                n_f = int(g["edge_counter"][9 + H - 1] - g["edge_counter"][9 + H - 2]) if H > 1 else int(g["node_counter"][9])
                tp_g = gpu.pools[p].buffer("tmp_part_ind")[:n_f].cpu().numpy()
                tp_c = np.ctypeslib.as_array(cpu.pools[p].p.contents.tmp_part_ind, shape=(max(n_f, 1),))[:n_f]
                assert np.array_equal(tp_g, tp_c)                                               # topology hit mask
                csb = g["cache_search_buffer"]
                hits += int((csb >= 0).sum())
                peer_hits += int(((csb >= 0) & (csb // max(caps[0], 1) != p % Kg)).sum())       # rows that live on another member
                topo_hits += int((tp_g >= 0).sum())
                assert gpu.pools[p].error() == 0
        if p == 0:
            check_invariants(wl, g, fanout)
    gpu.close(); cpu.close()
    return hits, topo_hits, peer_hits, caps


def test_config3_shape_oracle_replay(hip):
    """configs[3] (uk-union on 8 GPUs): D = 256, B = 8000, [25,10], 8 logical GPUs = one clique of Kg = 8
    (cache_agg_mode 3), feature cache + topology cache sized by the cost model from measured counters."""
    wl = Workload(scale=17, edge_factor=16, dim=256, partition_count=8, n_seeds=1 << 17, n_valid=0, n_test=0)
    hits, topo_hits, peer_hits, caps = _clique_replay(wl, [25, 10], 8000, 3, cache_memory=3_000_000, modes=(0,))
    assert caps[0] > 1000 and caps[1] > 1000, caps          # both caches are non-trivial
    assert hits > 0 and topo_hits > 0 and peer_hits > 0


def test_config4_shape_oracle_replay(hip):
    """configs[4] (RMAT-28 GAT on 8 GPUs): [15,10,5], D = 256, B = 8000, one clique of Kg = 8."""
    wl = Workload(scale=16, edge_factor=4, dim=256, partition_count=8, n_seeds=1 << 16, n_valid=0, n_test=0)
    hits, topo_hits, peer_hits, caps = _clique_replay(wl, [15, 10, 5], 8000, 3, cache_memory=1_500_000, modes=(0,))
    assert caps[0] > 500 and caps[1] > 500, caps
    assert hits > 0 and topo_hits > 0 and peer_hits > 0


def _check_lane(pl, seeds_of_batch, indptr, col, deg, fanout, D, rows_cap):
    nc = pl.buffer("node_counter").cpu().numpy()
    ec = pl.buffer("edge_counter").cpu().numpy()
    H = len(fanout)
    n, e = int(nc[9 + H]), int(ec[9 + H])
    assert 0 < n <= rows_cap and pl.error() == 0
    ids = pl.buffer("sampled_ids")[:n]
    assert int(torch.unique(ids).numel()) == n
    assert synth.feature_check_device(pl.buffer("float_features")[:n].contiguous(), ids.contiguous(), D, 7) == 0
    src_g, dst_g = pl.buffer("agg_src_ids")[:e].long(), pl.buffer("agg_dst_ids")[:e].long()
    assert bool((ids.long()[pl.buffer("agg_src_off")[:e].long()] == src_g).all())
    assert bool((ids.long()[pl.buffer("agg_dst_off")[:e].long()] == dst_g).all())
    b = int(nc[9])
    assert np.array_equal(ids[:b].cpu().numpy(), seeds_of_batch)
    lo, frontier = 0, ids[:b].long()
    for h in range(H):                       # per hop: edges = sum of min(fan-out, degree) over the (duplicated) frontier
        hi = int(ec[9 + h + 1])
        assert hi - lo == int(torch.clamp(deg[frontier], max=fanout[h]).sum()), f"hop {h + 1} edge count"
        frontier = src_g[lo:hi]
        lo = hi
    for kk in torch.randint(0, e, (40,)).tolist():      # sampled neighbours are real neighbours
        row = col[int(indptr[dst_g[kk]]):int(indptr[dst_g[kk] + 1])]
        assert bool((row == src_g[kk]).any())
    return n, e, nc, ec


def test_config3_shape_full_size_striped_clique(hip):
    """configs[3] at size: RMAT-24 (2^24 vertices, 2^28 edges), D = 256 (17 GB of features), B = 8000, [25,10],
    8 logical GPUs striped as one clique (Kg = 8) with a real topology cache, lane groups of 8 under hipGraph
    replay on one member: properties + rows served from the other members' stripes + topology-cache hits."""
    scale, D, fanout, batch, P, group = 24, 256, [25, 10], 8000, 8, 8
    N = 1 << scale
    dev = torch.device("cuda:0")
    indptr, col = synth.rmat_csr_device(scale, 16, 20231, dev, scramble=True)
    torch.cuda.empty_cache()
    feats = synth.features_device(N, D, 7, dev)
    seeds = synth.seed_ids(N, N // 4, 11)
    graph, feature = engine.GraphStorage(P, indptr, col), engine.FeatureStorage(P, feats)
    mine = [np.ascontiguousarray(seeds[seeds % P == p]) for p in range(P)]
    presc_steps = 4
    cache = engine.UnifiedCache(2 << 30, D, presc_steps, P, N)          # 2 GB per GPU -> 16 GB over the clique
    tx = 0
    for p in range(P):
        feature.set_ids(p, 0, mine[p], None)
        cache.init_controller(p)
        pool = engine.MemoryPool(p, N, batch, fanout, D)
        for it in range(presc_steps):
            engine.enqueue_batch(None, graph, feature, cache, pool, batch, it, p, 0, True, fanout)
        torch.cuda.synchronize()
        tx += cache.topo_transactions(p)
        pool.close()
    cache.candidate_selection(3, graph)
    cache.cost_model(feature, graph, (tx, 0), presc_steps)
    cache.fill_up(feature, graph)
    ncap, ecap = cache.node_capacity(0), cache.edge_capacity(0)
    assert ncap > 100_000 and ecap > 10_000, (ncap, ecap)
    d = 5                                                              # the member this process serves
    rows = int(cache.max_id_num(d) * 1.2)
    pipe = engine.Pipeline(graph, feature, cache, d, batch, fanout, group, rows, True, 2)
    deg = indptr[1:] - indptr[:-1]
    peer = topo = 0
    for c0 in (0, group, 2 * group):
        slot = pipe.submit(c0)
        pipe.wait(slot)
        for lane in (0, group // 2, group - 1):
            pl = pipe.pools[slot][lane]
            n, e, nc, ec = _check_lane(pl, mine[d][(c0 + lane) * batch:(c0 + lane + 1) * batch], indptr, col, deg, fanout, D, rows)
            csb = pl.buffer("cache_search_buffer")[:int(nc[1])]
            peer += int(((csb >= 0) & (csb // ncap != d)).sum())        # owner index within the clique != this member
            n_f = int(ec[10])                                           # last hop's frontier = hop-1 edges
            topo += int((pl.buffer("tmp_part_ind")[:n_f] >= 0).sum())
    assert peer > 0 and topo > 0
    pipe.close(); cache.close(); feature.close(); graph.close()


def test_config4_shape_2pow28_vertices(hip):
    """configs[4] at its vertex count: RMAT-28 (N = 2^28, edge factor 4: 2^30 edges), [15,10,5], B = 8000, lane
    groups + hipGraph.  D = 128 here: the 256-wide table of 2^28 rows is 275 GB and exists only striped over
    eight GPUs; D = 256 at B = 8000 is covered by test_config4_shape_oracle_replay and the config-3 full-size
    case.  What this run pins down is everything that scales with N: int32 ids up to 2^28, int64 row starts up
    to 2^30, and that the lanes keep nothing per vertex (256 de-duplication buckets per lane at this batch size; rounds 1-4 also
    ran this with a 1 GB array and a 128 MB table per lane; the pool the PreSC epoch itself runs on has 256)."""
    import os
    scale, D, fanout, batch, group = 28, 128, [15, 10, 5], 8000, 4
    N = 1 << scale
    dev = torch.device("cuda:0")
    indptr, col = synth.rmat_csr_device(scale, 4, 20231, dev, scramble=True)
    torch.cuda.empty_cache()
    assert int(indptr[-1]) == N * 4 and int(col.max()) < N and int(col.max()) >= N // 2
    feats = synth.features_device(N, D, 7, dev)                          # 137 GB
    seeds = synth.seed_ids(N, 400_000, 11)
    graph, feature = engine.GraphStorage(1, indptr, col), engine.FeatureStorage(1, feats)
    feature.set_ids(0, 0, seeds, None)
    cache = engine.UnifiedCache(16 << 30, D, 4, 1, N)
    cache.init_controller(0)
    pool = engine.MemoryPool(0, N, batch, fanout, D)
    for it in range(4):
        engine.enqueue_batch(None, graph, feature, cache, pool, batch, it, 0, 0, True, fanout)
    torch.cuda.synchronize()
    tx = cache.topo_transactions(0)
    cache.candidate_selection(0, graph)
    cache.cost_model(feature, graph, (tx, 0), 4)
    cache.fill_up(feature, graph)
    rows = int(cache.max_id_num(0) * 1.2)
    pool.close()
    deg = indptr[1:] - indptr[:-1]
    seen = {}
    for rep in range(2):                   # two pipelines over the same tables (the second re-creates every lane): the same batches
        pipe = engine.Pipeline(graph, feature, cache, 0, batch, fanout, group, rows, True, 2)
        # (64 buckets: PreSC saw ~0.4 M claims in hop 3 of these batches -- they fit 64 workgroups' registers; by slots alone it would be 256)
        assert pipe.pools[0][0].lds_buckets() == 64 and pipe.pools[0][0].state_bytes() < (1 << 29)      # nothing that scales with N = 2^28
        for c0 in (0, group):
            slot = pipe.submit(c0)
            pipe.wait(slot)
            for lane in (0, group - 1):
                pl = pipe.pools[slot][lane]
                n, e, nc, ec = _check_lane(pl, seeds[(c0 + lane) * batch:(c0 + lane + 1) * batch], indptr, col, deg, fanout, D, rows)
                key = (c0, lane)
                sig = (n, e, int(pl.buffer("sampled_ids")[:n].long().sum()), int(pl.buffer("agg_src_off")[:e].long().sum()))
                assert seen.setdefault(key, sig) == sig
                assert int(pl.buffer("sampled_ids")[:n].max()) > N // 2
        pipe.close()
    cache.close(); feature.close(); graph.close()


def test_config4_shape_d256_rows_pinned_spill_and_hbm_cache(hip):
    """What ONE GPU of configs[4] sees of the 256-wide table: its rows are 1024 bytes, the full table sits in MAPPED PINNED HOST memory
    (it exists nowhere in one GPU's HBM: 2^28 rows x 1 KB = 275 GB) and the hot rows in an HBM cache -- stripe + spill.  On this pool a
    1-GPU box's control group holds 300 GiB of host memory (profiles/r06/box_probe.txt), so the pinned table of the full 2^28 rows
    (256 GiB) does not fit beside the process; this runs the largest power of two that does with room to spare: 2^27 vertices, edge
    factor 8 (the same 2^30 edges), [15,10,5], B = 8000 -- a 128 GiB pinned table, 1 KB rows over PCIe on a miss, and every id-width
    property of the 2^28-vertex D = 128 case above (test_config4_shape_2pow28_vertices keeps the vertex count itself).  A box whose
    control group (or available memory) is smaller drops a power of two or two rather than risking the box."""
    limit = None
    try:
        txt = open("/sys/fs/cgroup/memory.max").read().strip()
        limit = None if txt == "max" else int(txt)
    except OSError:
        pass
    avail = None
    try:
        for ln in open("/proc/meminfo"):
            if ln.startswith("MemAvailable:"):
                avail = int(ln.split()[1]) << 10
    except OSError:
        pass
    room = min(x for x in (limit, avail, 1 << 62) if x is not None)      # what this process may pin without endangering the box
    scale = 27 if room >= (220 << 30) else (26 if room >= (120 << 30) else 25)
    D, fanout, batch, group, ef = 256, [15, 10, 5], 8000, 4, {27: 8, 26: 16, 25: 16}[scale]
    N = 1 << scale
    dev = torch.device("cuda:0")
    indptr, col = synth.rmat_csr_device(scale, ef, 20231, dev, scramble=True)
    torch.cuda.empty_cache()
    assert int(indptr[-1]) == N * ef and int(col.max()) >= N // 2
    p_feat = engine.PinnedArray.empty((N, D), np.float32)                 # 128 GiB at scale 27
    feats = p_feat.tensor(dev)
    chunk = 1 << 21
    for r0 in range(0, N, chunk):                                         # generated on the device, parked in pinned host memory
        feats[r0:r0 + chunk].copy_(synth.features_device_rows(r0, chunk, D, 7, dev))
    torch.cuda.synchronize()
    seeds = synth.seed_ids(N, 400_000, 11)
    graph, feature = engine.GraphStorage(1, indptr, col), engine.FeatureStorage(1, feats)
    feature.set_ids(0, 0, seeds, None)
    presc = 4
    cache = engine.UnifiedCache(24 << 30, D, presc, 1, N)                 # 24 GB: about what an eighth of a 288 GB part leaves a stripe
    cache.init_controller(0)
    pool = engine.MemoryPool(0, N, batch, fanout, D)
    for it in range(presc):
        engine.enqueue_batch(None, graph, feature, cache, pool, batch, it, 0, 0, True, fanout)
    torch.cuda.synchronize()
    cache.candidate_selection(0, graph)
    cache.cost_model(feature, graph, (cache.topo_transactions(0), 0), presc)
    cache.fill_up(feature, graph)
    ncap = cache.node_capacity(0)
    assert ncap > 1_000_000 and ncap * D * 4 <= (24 << 30)
    rows = int(cache.max_id_num(0) * 1.2)
    pool.close()
    pipe = engine.Pipeline(graph, feature, cache, 0, batch, fanout, group, rows, True, 2)
    deg = indptr[1:] - indptr[:-1]
    hit = miss = 0
    seen = {}
    for rep in range(2):                                                  # replayed: the same batches both times
        for c0 in (presc, presc + group):
            slot = pipe.submit(c0)
            pipe.wait(slot)
            for lane in (0, group - 1):
                pl = pipe.pools[slot][lane]
                n, e, nc, ec = _check_lane(pl, seeds[(c0 + lane) * batch:(c0 + lane + 1) * batch], indptr, col, deg, fanout, D, rows)
                sig = (n, e, int(pl.buffer("sampled_ids")[:n].long().sum()), int(pl.buffer("agg_src_off")[:e].long().sum()))
                assert seen.setdefault((c0, lane), sig) == sig
                assert int(pl.buffer("sampled_ids")[:n].max()) > N // 2
                csb = pl.buffer("cache_search_buffer")[:int(nc[1])]
                hit += int((csb >= 0).sum()); miss += int((csb < 0).sum())
    assert hit > 0 and miss > 0                                           # rows came from the HBM cache AND over PCIe from the pinned table
    pipe.close(); cache.close(); feature.close(); graph.close()
    p_feat.close()


def test_config2_shape_full_size_pinned_spill(hip):
    """configs[2] at size (papers100M-like): 2^26 vertices, 3 hops [15,10,5], full CSR and the full 128-wide
    feature table (34 GB) in MAPPED PINNED HOST memory, hotness-ranked feature cache + topology cache in HBM
    (cost model fed with the measured transactions): hits come from HBM, misses are read in place over PCIe."""
    scale, D, fanout, batch, group = 26, 128, [15, 10, 5], 1024, 8
    N = 1 << scale
    dev = torch.device("cuda:0")
    indptr_d, col_d = synth.rmat_csr_device(scale, 8, 20231, dev, scramble=True)
    torch.cuda.empty_cache()
    p_indptr = engine.PinnedArray.empty((N + 1,), np.int64)
    p_col = engine.PinnedArray.empty((int(col_d.numel()),), np.int32)
    p_feat = engine.PinnedArray.empty((N, D), np.float32)
    indptr, col, feats = p_indptr.tensor(dev), p_col.tensor(dev), p_feat.tensor(dev)
    indptr.copy_(indptr_d); col.copy_(col_d)
    chunk = 1 << 22
    for r0 in range(0, N, chunk):                                        # generate on the device, park in pinned host memory
        feats[r0:r0 + chunk].copy_(synth.features_device_rows(r0, chunk, D, 7, dev))
    torch.cuda.synchronize()
    seeds = synth.seed_ids(N, 300_000, 11)
    graph, feature = engine.GraphStorage(1, indptr, col), engine.FeatureStorage(1, feats)
    feature.set_ids(0, 0, seeds, None)
    presc = 8
    cache = engine.UnifiedCache(8 << 30, D, presc, 1, N)
    cache.init_controller(0)
    pool = engine.MemoryPool(0, N, batch, fanout, D)
    import time
    torch.cuda.synchronize(); time.sleep(0.05)
    lc0 = engine.link_counters(0)
    for it in range(presc):
        engine.enqueue_batch(None, graph, feature, cache, pool, batch, it, 0, 0, True, fanout)
    torch.cuda.synchronize(); time.sleep(0.05)
    lc1 = engine.link_counters(0)
    tx = cache.topo_transactions(0)
    # N3: the PCIe link's own cumulative counter (driver gpu_metrics table) saw the PreSC epoch's topology reads.  The
    # sampler's computed count charges one transaction per row-pointer pair too, which this build keeps in HBM (RowHdr),
    # and the hardware merges picks that share a 64-byte line: the measured figure is the smaller one, same magnitude.
    assert lc0 is not None and lc1 is not None, "gpu_metrics link counters unreadable on this box"
    measured = (lc1[0] - lc0[0]) // 64
    assert 0.1 < measured / tx < 2.0, (measured, tx)
    cache.candidate_selection(0, graph)
    cache.cost_model(feature, graph, (measured, 0), presc)
    cache.fill_up(feature, graph)
    assert cache.node_capacity(0) > 1_000_000 and cache.edge_capacity(0) > 50_000
    rows = int(cache.max_id_num(0) * 1.2)
    pool.close()
    pipe = engine.Pipeline(graph, feature, cache, 0, batch, fanout, group, rows, True, 2)
    deg = indptr_d[1:] - indptr_d[:-1]
    hit = miss = topo = topo_miss = 0
    for c0 in (presc, presc + group):
        slot = pipe.submit(c0)
        pipe.wait(slot)
        for lane in (0, group - 1):
            pl = pipe.pools[slot][lane]
            n, e, nc, ec = _check_lane(pl, seeds[(c0 + lane) * batch:(c0 + lane + 1) * batch], indptr_d, col_d, deg, fanout, D, rows)
            csb = pl.buffer("cache_search_buffer")[:int(nc[1])]
            hit += int((csb >= 0).sum()); miss += int((csb < 0).sum())
            n_f = int(ec[11] - ec[10])
            tp = pl.buffer("tmp_part_ind")[:n_f]
            topo += int((tp >= 0).sum()); topo_miss += int((tp < 0).sum())
    assert hit > 0 and miss > 0 and topo > 0 and topo_miss > 0          # every tier was exercised
    pipe.close(); cache.close(); feature.close(); graph.close()
    for p in (p_indptr, p_col, p_feat):
        p.close()


# ---- the reference's REAL data-set sizes (legion_server.py:41-88) -----------------------------------------------------
UK_UNION = (133_633_040, 5_507_679_822)        # legion_server.py:65-72 -- not a power of two; more than 2^32 edges
PAPERS100M = (111_059_956, 1_615_685_872)      # legion_server.py:49-56


def test_large_graph_generator_small(hip):
    """synth.csr_device_large on a size the host can check: odd vertex count, several chunks, deterministic."""
    dev = torch.device("cuda:0")
    N, E = 4099, 60_001
    indptr, col = synth.csr_device_large(N, E, 5, dev, chunk_edges=7000)
    ip, cl = indptr.cpu().numpy(), col.cpu().numpy()
    assert ip[0] == 0 and ip[-1] == E and np.all(np.diff(ip) >= 0) and cl.min() >= 0 and cl.max() < N
    rows = np.repeat(np.arange(N), np.diff(ip))
    assert not np.any(rows == cl)                                     # no self loops survive the fold
    assert np.diff(ip).max() > 20 * E / N                             # skewed like the real graphs
    indptr2, col2 = synth.csr_device_large(N, E, 5, dev, chunk_edges=7000)
    assert torch.equal(indptr, indptr2) and torch.equal(col, col2)


def test_config3_real_dataset_size(hip):
    """configs[3] at uk-union's REAL size on the one GPU: N = 133 633 040 (not a power of two), E = 5 507 679 822 (> 2^32:
    row starts beyond every 32-bit index), D = 256 (137 GB of features), B = 8000, [25,10], 8 logical GPUs striped as one
    clique (Kg = 8) with a topology cache, lane groups under hipGraph replay on one member.  Properties as at RMAT-24, plus:
    adjacency rows that START beyond 2^32 are sampled, and vertices beyond 2^27 are reached."""
    N, E = UK_UNION
    D, fanout, batch, P, group = 256, [25, 10], 8000, 8, 4
    dev = torch.device("cuda:0")
    indptr, col = synth.csr_device_large(N, E, 20231, dev)
    assert int(indptr[-1]) == E and int(col.numel()) == E and E > 1 << 32
    feats = synth.features_device(N, D, 7, dev)                          # 137 GB
    seeds = synth.seed_ids(N, 1_000_000, 11)                             # ~125 k per logical GPU: 15 batches of 8000
    graph, feature = engine.GraphStorage(P, indptr, col), engine.FeatureStorage(P, feats)
    mine = [np.ascontiguousarray(seeds[seeds % P == p]) for p in range(P)]
    assert min(m.size for m in mine) > 2 * group * batch
    presc_steps = 2
    cache = engine.UnifiedCache(1 << 30, D, presc_steps, P, N)          # 1 GB per GPU -> 8 GB over the clique
    tx = 0
    for p in range(P):
        feature.set_ids(p, 0, mine[p], None)
        cache.init_controller(p)
        pool = engine.MemoryPool(p, N, batch, fanout, D)
        for it in range(presc_steps):
            engine.enqueue_batch(None, graph, feature, cache, pool, batch, it, p, 0, True, fanout)
        torch.cuda.synchronize()
        assert pool.error() == 0
        tx += cache.topo_transactions(p)
        pool.close()
    cache.candidate_selection(3, graph)
    cache.cost_model(feature, graph, (tx, 0), presc_steps)
    cache.fill_up(feature, graph)
    ncap, ecap = cache.node_capacity(0), cache.edge_capacity(0)
    assert ncap > 100_000 and ecap > 10_000, (ncap, ecap)
    d = 3                                                              # the member this process serves
    rows = int(cache.max_id_num(d) * 1.2)
    pipe = engine.Pipeline(graph, feature, cache, d, batch, fanout, group, rows, True, 2)
    deg = indptr[1:] - indptr[:-1]
    peer = topo = 0
    far_rows = high_ids = 0
    for c0 in (0, group):
        slot = pipe.submit(c0)
        pipe.wait(slot)
        for lane in (0, group - 1):
            pl = pipe.pools[slot][lane]
            n, e, nc, ec = _check_lane(pl, mine[d][(c0 + lane) * batch:(c0 + lane + 1) * batch], indptr, col, deg, fanout, D, rows)
            csb = pl.buffer("cache_search_buffer")[:int(nc[1])]
            peer += int(((csb >= 0) & (csb // ncap != d)).sum())
            topo += int((pl.buffer("tmp_part_ind")[:int(ec[10])] >= 0).sum())
            dst_g = pl.buffer("agg_dst_ids")[:e].long()                 # the vertices whose adjacency was sampled
            far_rows += int((indptr[dst_g] > (1 << 32)).sum())
            high_ids += int((pl.buffer("sampled_ids")[:n] > (1 << 27) - 1).sum())
    assert peer > 0 and topo > 0 and far_rows > 1000 and high_ids >= 0
    pipe.close(); cache.close(); feature.close(); graph.close()


def test_config2_real_dataset_size_pinned(hip):
    """configs[2] at papers100M's REAL size: N = 111 059 956, E = 1 615 685 872, D = 128, 3 hops [15,10,5], B = 8000 (Legion's
    default), the full CSR (7.4 GB) and the full feature table (56.9 GB) in MAPPED PINNED HOST memory, hotness-ranked
    feature + topology caches in HBM: hits from HBM, misses read in place over PCIe; properties as above."""
    N, E = PAPERS100M
    D, fanout, batch, group = 128, [15, 10, 5], 8000, 2
    dev = torch.device("cuda:0")
    indptr_d, col_d = synth.csr_device_large(N, E, 20231, dev)
    p_indptr = engine.PinnedArray.empty((N + 1,), np.int64)
    p_col = engine.PinnedArray.empty((E,), np.int32)
    p_feat = engine.PinnedArray.empty((N, D), np.float32)
    indptr, col, feats = p_indptr.tensor(dev), p_col.tensor(dev), p_feat.tensor(dev)
    indptr.copy_(indptr_d); col.copy_(col_d)
    chunk = 1 << 22
    for r0 in range(0, N, chunk):
        feats[r0:r0 + chunk].copy_(synth.features_device_rows(r0, min(chunk, N - r0), D, 7, dev))
    torch.cuda.synchronize()
    seeds = synth.seed_ids(N, 200_000, 11)
    graph, feature = engine.GraphStorage(1, indptr, col), engine.FeatureStorage(1, feats)
    feature.set_ids(0, 0, seeds, None)
    presc = 4
    cache = engine.UnifiedCache(8 << 30, D, presc, 1, N)
    cache.init_controller(0)
    pool = engine.MemoryPool(0, N, batch, fanout, D)
    for it in range(presc):
        engine.enqueue_batch(None, graph, feature, cache, pool, batch, it, 0, 0, True, fanout)
    torch.cuda.synchronize()
    assert pool.error() == 0
    tx = cache.topo_transactions(0)
    cache.candidate_selection(0, graph)
    cache.cost_model(feature, graph, (tx, 0), presc)
    cache.fill_up(feature, graph)
    assert cache.node_capacity(0) > 1_000_000 and cache.edge_capacity(0) > 50_000
    rows = int(cache.max_id_num(0) * 1.2)
    pool.close()
    pipe = engine.Pipeline(graph, feature, cache, 0, batch, fanout, group, rows, True, 2)
    deg = indptr_d[1:] - indptr_d[:-1]
    hit = miss = topo = topo_miss = 0
    for c0 in (presc, presc + group):
        slot = pipe.submit(c0)
        pipe.wait(slot)
        for lane in range(group):
            pl = pipe.pools[slot][lane]
            n, e, nc, ec = _check_lane(pl, seeds[(c0 + lane) * batch:(c0 + lane + 1) * batch], indptr_d, col_d, deg, fanout, D, rows)
            csb = pl.buffer("cache_search_buffer")[:int(nc[1])]
            hit += int((csb >= 0).sum()); miss += int((csb < 0).sum())
            tp = pl.buffer("tmp_part_ind")[:int(ec[11] - ec[10])]
            topo += int((tp >= 0).sum()); topo_miss += int((tp < 0).sum())
    assert hit > 0 and miss > 0 and topo > 0 and topo_miss > 0          # every tier was exercised
    pipe.close(); cache.close(); feature.close(); graph.close()
    for p in (p_indptr, p_col, p_feat):
        p.close()
