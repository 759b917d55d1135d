"""Parity of the HIP path (through the C ABI) with the oracle: bit-exact ids, positions, counters,
hit masks; byte-identical gathered rows."""
import numpy as np
import pytest
import torch

from tests.gpu_harness import CpuSide, GpuSide
from tests.helpers import Workload, check_invariants, compare_batches

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("scale,ef,fanout,batch,dim", [
    (10, 8, [25, 10], 64, 16),
    (12, 16, [25, 10], 256, 100),
    (12, 4, [15, 10, 5], 128, 128),
    (11, 8, [3], 500, 7),
    (13, 8, [10, 10], 1024, 256),
    (10, 2, [1, 1, 1, 1], 32, 6),
    (10, 2, [2, 2, 2, 2, 2], 32, 6),            # five and six hops: counter words 14 and 15 are the block's last
    (11, 4, [3, 2, 2, 2, 2, 2], 24, 8),         # (SS/engine/operator_impl.cu:67,81-82; hop_num <= 6)
])
def test_serve_batches_no_cache(hip, buckets, scale, ef, fanout, batch, dim):
    wl = Workload(scale=scale, edge_factor=ef, dim=dim)
    gpu, cpu = GpuSide(wl, batch, fanout), CpuSide(wl, batch, fanout)
    assert gpu.pools[0].lds_buckets() == buckets
    n_train = wl.sets[(0, 0)][0].size
    for mode, counters in ((0, range(min(3, (n_train - 1) // batch))), (1, range(2)), (2, range(1))):
        bs = batch if mode == 0 else min(batch, 100)
        for counter in counters:
            g, c = gpu.run(0, counter, mode, batch_size=bs), cpu.run(0, counter, mode, batch_size=bs)
            compare_batches(g, c, f"mode {mode} batch {counter}: ")
            assert np.all(g["cache_search_buffer"] == -2)          # nothing cached yet: all misses
            check_invariants(wl, g, fanout)
    gpu.close(); cpu.close()


def test_partial_last_batch_and_empty(hip, buckets):
    wl = Workload(scale=9, edge_factor=8, dim=8, n_seeds=100, n_valid=37, n_test=5)
    gpu, cpu = GpuSide(wl, 16, [4, 3]), CpuSide(wl, 16, [4, 3])
    for counter in (1, 2, 3):       # 37 ids, batch 16: full, partial (5, read at the reference's quirky offset), empty
        g, c = gpu.run(0, counter, 1), cpu.run(0, counter, 1)
        compare_batches(g, c, f"valid batch {counter}: ")
    gpu.close(); cpu.close()


@pytest.mark.parametrize("P,mode_bits,capacity", [(1, 0, (300, 200)), (2, 1, (150, 90)), (4, 2, (64, 33)),
                                                  (4, 1, (100, 50)), (8, 3, (40, 20)), (3, 0, (77, 10))])
def test_presc_cache_build_and_serve(hip, buckets, col_slots, P, mode_bits, capacity):
    """PreSC epoch -> hotness -> order -> maps/fills -> serving with hits, on P logical GPUs striped
    over cliques of 2^mode_bits (logical GPUs share the physical one on a 1-GPU box).  Once with the gather looking every
    row's cache slot up in node_map, once with the slots carried from the sampler (column slots): identical."""
    wl = Workload(scale=11, edge_factor=8, dim=32, partition_count=P, n_seeds=1200)
    fanout, batch = [5, 4], 64
    gpu, cpu = GpuSide(wl, batch, fanout), CpuSide(wl, batch, fanout)
    steps = min((wl.sets[(p, 0)][0].size - 1) // batch for p in range(P))
    assert steps >= 1
    for p in range(P):
        for it in range(steps):
            g, c = gpu.run(p, it, 0, is_presc=True), cpu.run(p, it, 0, is_presc=True)
            compare_batches(g, c, f"presc gpu {p} it {it}: ")
    for p in range(P):
        assert np.array_equal(gpu.cache.array("node_access_time", p).cpu().numpy().view(np.uint64), cpu.node_access[p])
        assert np.array_equal(gpu.cache.array("edge_access_time", p).cpu().numpy().view(np.uint64), cpu.edge_access[p])
        assert gpu.cache.max_id_num(p) == cpu.max_ids[p]
    gpu.cache.candidate_selection(mode_bits, gpu.graph)
    gpu.cache.set_capacity(*capacity)
    gpu.cache.fill_up(gpu.feature, gpu.graph)
    assert all(gpu.graph.column_slots(p) == col_slots for p in range(P))
    caches = cpu.build_cache(mode_bits, capacity=capacity)
    Kg = cpu.Kg
    for ki, oc in enumerate(caches):
        lead = ki * Kg
        assert np.array_equal(gpu.cache.array("QF", lead).cpu().numpy(), oc.arr("QF", np.int32))
        assert np.array_equal(gpu.cache.array("QT", lead).cpu().numpy(), oc.arr("QT", np.int32))
        assert np.array_equal(gpu.cache.array("AF", lead).cpu().numpy().view(np.uint64), oc.arr("AF", np.uint64))
        for j in range(Kg):
            assert np.array_equal(gpu.cache.array("node_map", lead + j).cpu().numpy(), oc.arr("node_map", np.int32))
            assert np.array_equal(gpu.cache.array("edge_index_map", lead + j).cpu().numpy(), oc.arr("edge_index_map", np.int8))
            assert np.array_equal(gpu.cache.array("edge_offset_map", lead + j).cpu().numpy(), oc.arr("edge_offset_map", np.int32))
    hits = 0
    for p in range(P):
        for mode in (0, 1):
            g, c = gpu.run(p, 0, mode), cpu.run(p, 0, mode)
            compare_batches(g, c, f"serve gpu {p} mode {mode}: ")
            assert np.array_equal(g["cache_search_buffer"], c["cache_search_buffer"])      # feature hit mask + slots
            H = len(fanout)                  # the last sampler pass covered the last hop's frontier
            n_f = int(g["edge_counter"][9 + H - 1] - g["edge_counter"][9 + H - 2]) if H > 1 else int(g["node_counter"][9])
            tp_g = gpu.pools[p].buffer("tmp_part_ind")[:n_f].cpu().numpy()
            tp_c = np.ctypeslib.as_array(cpu.pools[p].p.contents.tmp_part_ind, shape=(max(n_f, 1),))[:n_f]
            assert np.array_equal(tp_g, tp_c)                                               # topology hit mask
            # explicit FindTopo (owner + row offset) on the same frontier
            f_lo = int(g["edge_counter"][9 + H - 2]) if H > 1 else 0
            frontier = (g["agg_src_ids"][f_lo:f_lo + n_f] if H > 1 else g["sampled_ids"][:n_f]).astype(np.int32)
            ind, off = gpu.cache.find_topo(p, torch.from_numpy(frontier).cuda())
            torch.cuda.synchronize()
            to_c = np.ctypeslib.as_array(cpu.pools[p].p.contents.tmp_part_off, shape=(max(n_f, 1),))[:n_f]
            assert np.array_equal(ind.cpu().numpy(), tp_c) and np.array_equal(off.cpu().numpy(), to_c)
            hits += int((g["cache_search_buffer"] >= 0).sum())
    assert hits > 0
    gpu.close(); cpu.close()


@pytest.mark.parametrize("cache_memory,counters", [(200_000, (0, 0)), (1_000_000, (0, 0)),
                                                   (400_000, (50_000, 70_000)), (3_000_000, (9, 9)),
                                                   (50_000_000, (0, 0))])
def test_cost_model_matches_oracle(hip, cache_memory, counters):
    wl = Workload(scale=12, edge_factor=8, dim=64, n_seeds=2000)
    fanout, batch = [10, 5], 128
    gpu, cpu = GpuSide(wl, batch, fanout, cache_memory=cache_memory), CpuSide(wl, batch, fanout)
    steps = (wl.sets[(0, 0)][0].size - 1) // batch
    for it in range(steps):
        gpu.run(0, it, 0, is_presc=True)
        cpu.run(0, it, 0, is_presc=True)
    gpu.cache.candidate_selection(0, gpu.graph)
    gpu.cache.cost_model(gpu.feature, gpu.graph, counters, steps)
    oc = cpu.build_cache(0, cache_memory=cache_memory, train_step=steps, counters=counters)[0]
    assert (gpu.cache.node_capacity(0), gpu.cache.edge_capacity(0)) == (oc.node_capacity, oc.edge_capacity)
    gpu.cache.fill_up(gpu.feature, gpu.graph)
    g, c = gpu.run(0, 0, 0), cpu.run(0, 0, 0)
    compare_batches(g, c, "serve after cost model: ")
    assert np.array_equal(g["cache_search_buffer"], c["cache_search_buffer"])
    gpu.close(); cpu.close()


def expected_topo_transactions(wl, batch, fanout):
    """What the sampler counts for one PreSC batch (legion_hip.h): per real frontier row one 64-byte
    transaction for the row-pointer pair plus min(fan-out, ceil(4*deg/64)) for the picks."""
    nc, ec = batch["node_counter"], batch["edge_counter"]
    deg = np.diff(wl.indptr)
    total = 0
    for h, count in enumerate(fanout):
        if h == 0:
            frontier = batch["sampled_ids"][:int(nc[9])]
        else:
            frontier = batch["agg_src_ids"][int(ec[9 + h - 1]):int(ec[9 + h])]
        frontier = frontier[frontier >= 0]
        total += int(np.sum(1 + np.minimum(count, (deg[frontier] * 4 + 63) // 64)))
    return total


def test_presc_topology_transactions_feed_cost_model(hip):
    """The sampler counts, during PreSC, the 64-byte transactions its topology reads amount to (the PCM
    counter of the paper, zero in v2).  The count matches the same sum taken over the oracle's batches
    (three hops, partial last batch), and the cost model fed with it picks the capacities the oracle's
    cost model picks for the same counters -- with a topology share this time."""
    wl = Workload(scale=12, edge_factor=8, dim=64, n_seeds=1000)
    fanout, batch = [10, 5, 3], 128
    cache_memory = 600_000
    gpu, cpu = GpuSide(wl, batch, fanout, cache_memory=cache_memory), CpuSide(wl, batch, fanout)
    steps = (wl.sets[(0, 0)][0].size + batch - 1) // batch        # including the partial last batch
    want = 0
    for it in range(steps):
        g, c = gpu.run(0, it, 0, is_presc=True), cpu.run(0, it, 0, is_presc=True)
        compare_batches(g, c, f"presc batch {it}: ")
        want += expected_topo_transactions(wl, c, fanout)
    got = gpu.cache.topo_transactions(0)
    assert got == want and got > 0
    gpu.cache.candidate_selection(0, gpu.graph)
    gpu.cache.cost_model(gpu.feature, gpu.graph, (got, 0), steps)
    oc = cpu.build_cache(0, cache_memory=cache_memory, train_step=steps, counters=(got, 0))[0]
    assert (gpu.cache.node_capacity(0), gpu.cache.edge_capacity(0)) == (oc.node_capacity, oc.edge_capacity)
    gpu.cache.cost_model(gpu.feature, gpu.graph, (0, 0), steps)
    v2_caps = (gpu.cache.node_capacity(0), gpu.cache.edge_capacity(0))
    gpu.cache.cost_model(gpu.feature, gpu.graph, (got, 0), steps)
    assert gpu.cache.edge_capacity(0) >= v2_caps[1]            # measured counters never shrink the topology share
    gpu.cache.fill_up(gpu.feature, gpu.graph)
    g, c = gpu.run(0, 0, 0), cpu.run(0, 0, 0)
    compare_batches(g, c, "serve after measured cost model: ")
    gpu.close(); cpu.close()


@pytest.mark.parametrize("chunk_mb", [None, "64", "0"], ids=["2MB-chunks", "64MB-chunks", "plain"])
def test_pipeline_lanes_in_a_scattered_arena(hip, monkeypatch, chunk_mb):
    """arena=True: every lane's trainer-visible arrays live in ONE arena built from physical chunks mapped in shuffled order (HIP
    virtual memory management, storage.hip d_alloc_scattered; LEGION_ARENA_SCATTER_MB=0: one plain allocation).  Same batches as
    the oracle's under graph replay and the weave, twice (the second pipeline re-uses what the first one released)."""
    from legion_amd import engine
    if chunk_mb is not None:
        monkeypatch.setenv("LEGION_ARENA_SCATTER_MB", chunk_mb)
    wl = Workload(scale=11, edge_factor=8, dim=32, n_seeds=700)
    fanout, batch, group = [6, 3], 64, 4
    gpu, cpu = GpuSide(wl, batch, fanout), CpuSide(wl, batch, fanout)
    for rep in range(2):
        pipe = engine.Pipeline(gpu.graph, gpu.feature, gpu.cache, 0, batch, fanout, group, gpu.pools[0].num_ids, True, 2, weave=True, arena=True)
        for gi in range(3):
            sl = pipe.submit(gi * group, 0)
            pipe.wait(sl)
            for lane in range(group):
                compare_batches(engine.read_batch(pipe.pools[sl][lane]), cpu.run(0, gi * group + lane, 0),
                                f"scattered arena ({chunk_mb}) rep {rep} batch {gi * group + lane}: ")
        pipe.close()
    gpu.close(); cpu.close()


@pytest.mark.parametrize("group,slots,use_graph,weave", [(1, 1, True, False), (3, 2, True, False), (4, 2, False, False),
                                                         (2, 3, True, False), (8, 2, True, False),
                                                         (3, 2, True, True), (4, 2, False, True), (2, 3, True, True),
                                                         (1, 1, True, True), (8, 2, True, True)])
def test_pipeline_groups_and_graph_replay(hip, buckets, group, slots, use_graph, weave):
    """Grouped launches (grid.y = lanes) + hipGraph replay produce exactly the batches the one-lane
    eager path does: every batch of a short run -- including the clamped last batch and the empty
    batches past the end of the set, whose sizes are computed on the device -- is compared with the
    oracle.  weave = the head of a group (seeds + every hop but the last) and its rest as two graphs on two streams."""
    from legion_amd import engine
    wl = Workload(scale=11, edge_factor=8, dim=32, n_seeds=700)
    fanout, batch = [6, 3], 64
    gpu, cpu = GpuSide(wl, batch, fanout), CpuSide(wl, batch, fanout)
    steps = (wl.sets[(0, 0)][0].size - 1) // batch
    for it in range(steps):
        gpu.run(0, it, 0, is_presc=True); cpu.run(0, it, 0, is_presc=True)
    gpu.cache.candidate_selection(0, gpu.graph)
    gpu.cache.set_capacity(150, 80)
    gpu.cache.fill_up(gpu.feature, gpu.graph)
    cpu.build_cache(0, capacity=(150, 80))
    pipe = engine.Pipeline(gpu.graph, gpu.feature, gpu.cache, 0, batch, fanout, group, gpu.pools[0].num_ids, use_graph, slots,
                           weave=weave)
    n_batches = (wl.sets[(0, 0)][0].size + batch - 1) // batch     # the last one is partial
    n_groups = (n_batches + group - 1) // group                    # the last group may reach past the set
    for rep in range(2):                                          # the second epoch re-positions the device iteration
        pending = []
        for gi in range(n_groups):
            slot = pipe.submit(gi * group, 0)
            pending.append((gi, slot))
            if len(pending) == slots or gi == n_groups - 1:
                for gj, sl in pending:
                    pipe.wait(sl)
                    for lane in range(group):
                        got = engine.read_batch(pipe.pools[sl][lane])
                        want = cpu.run(0, gj * group + lane, 0)
                        compare_batches(got, want, f"group {group} slots {slots} rep {rep} batch {gj * group + lane}: ")
                pending = []
    # a different mode on the same slots uses its own graph
    sl = pipe.submit(0, 1)
    pipe.wait(sl)
    for lane in range(min(group, 2)):
        compare_batches(engine.read_batch(pipe.pools[sl][lane]), cpu.run(0, lane, 1), f"valid batch {lane}: ")
    pipe.close()
    gpu.close(); cpu.close()


@pytest.mark.parametrize("fanout,group,slots,use_graph,weave", [([2, 2, 2, 2, 2], 3, 2, True, True), ([3, 2, 2, 2, 2, 2], 4, 2, True, True),
                                                                ([3, 2, 2, 2, 2, 2], 2, 3, True, False), ([2, 2, 2, 2, 2, 2], 3, 2, False, True)])
def test_five_and_six_hops_through_groups_graph_replay_and_weave(hip, buckets, fanout, group, slots, use_graph, weave):
    """hop_num = 5 and 6 (the counter block allows no more: words 9 + hop_num = 14, 15 are its last, SS/engine/operator_impl.cu:67,81-82)
    through PreSC, the cost model, a cache with cached topology, lane groups, hipGraph replay and the weave -- five / six known-list
    generations, six gathers per group -- every batch of two epochs against the oracle."""
    from legion_amd import engine
    wl = Workload(scale=11, edge_factor=8, dim=20, n_seeds=500)
    batch = 48
    gpu, cpu = GpuSide(wl, batch, fanout, cache_memory=120_000), CpuSide(wl, batch, fanout)
    steps = (wl.sets[(0, 0)][0].size - 1) // batch
    for it in range(steps):
        compare_batches(gpu.run(0, it, 0, is_presc=True), cpu.run(0, it, 0, is_presc=True), f"presc {it}: ")
    gpu.cache.candidate_selection(0, gpu.graph)
    gpu.cache.cost_model(gpu.feature, gpu.graph, (0, 0), steps)
    oc = cpu.build_cache(0, cache_memory=120_000, train_step=steps)[0]
    assert (gpu.cache.node_capacity(0), gpu.cache.edge_capacity(0)) == (oc.node_capacity, oc.edge_capacity)
    gpu.cache.fill_up(gpu.feature, gpu.graph)
    g, c = gpu.run(0, 0, 0), cpu.run(0, 0, 0)
    compare_batches(g, c, "eager serve: ")
    H = len(fanout)
    assert int(g["node_counter"][8]) == H and g["node_counter"][9 + H] == g["sampled_ids"].size and g["edge_counter"][9 + H] == g["agg_src_off"].size
    pipe = engine.Pipeline(gpu.graph, gpu.feature, gpu.cache, 0, batch, fanout, group, gpu.pools[0].num_ids, use_graph, slots, weave=weave)
    n_batches = (wl.sets[(0, 0)][0].size + batch - 1) // batch
    n_groups = (n_batches + group - 1) // group
    for rep in range(2):
        pending = []
        for gi in range(n_groups):
            pending.append((gi, pipe.submit(gi * group, 0)))
            if len(pending) == slots or gi == n_groups - 1:
                for gj, sl in pending:
                    pipe.wait(sl)
                    for lane in range(group):
                        compare_batches(engine.read_batch(pipe.pools[sl][lane]), cpu.run(0, gj * group + lane, 0),
                                        f"{H} hops rep {rep} batch {gj * group + lane}: ")
                        assert pipe.pools[sl][lane].error() == 0
                pending = []
    pipe.close()
    gpu.close(); cpu.close()


@pytest.mark.parametrize("batch,fanout,group", [(1024, [25, 10], 128), (2000, [15, 10, 5], 16), (256, [6, 4, 3, 2], 24)])
def test_full_width_groups_against_the_oracle(hip, batch, fanout, group):
    """The headline's own geometry -- B = 1024, [25,10], lane groups wide enough to keep every XCD busy, the weave arrangement
    under hipGraph replay -- with EVERY lane of two consecutive groups compared with the oracle: the
    ticketed tiles and the decoupled look-back of compact_kernel, the winners' published positions and the software-pipelined
    gather under the concurrency they run with in the bench (250 super tiles per lane at hop 2, 128 lanes in flight).  The
    second shape has three hops in the 64-bucket class (staged placement, known lists across two hops), the third four hops."""
    from legion_amd import engine
    wl = Workload(scale=19, edge_factor=16, dim=16, n_seeds=2 * group * batch + batch)
    gpu, cpu = GpuSide(wl, batch, fanout), CpuSide(wl, batch, fanout)
    for it in range(2):
        gpu.run(0, it, 0, is_presc=True); cpu.run(0, it, 0, is_presc=True)
    gpu.cache.candidate_selection(0, gpu.graph)
    gpu.cache.set_capacity(20_000, 2_000)
    gpu.cache.fill_up(gpu.feature, gpu.graph)
    cpu.build_cache(0, capacity=(20_000, 2_000))
    pipe = engine.Pipeline(gpu.graph, gpu.feature, gpu.cache, 0, batch, fanout, group, gpu.pools[0].num_ids, True, 2, weave=True)
    slots = [pipe.submit(0, 0), pipe.submit(group, 0)]
    for gi, sl in enumerate(slots):
        pipe.wait(sl)
        for lane in range(group):
            compare_batches(engine.read_batch(pipe.pools[sl][lane]), cpu.run(0, gi * group + lane, 0), f"group {gi} lane {lane}: ")
            assert pipe.pools[sl][lane].error() == 0
    pipe.close()
    gpu.close(); cpu.close()


def test_regather_last_is_the_groups_own_last_gather(hip, col_slots):
    """legion_pipeline_regather_last (bench.py's roofline.alone / roofline.cold): launching a group's last gather again leaves every
    lane exactly as it was; after the last hop's ids of the lanes were replaced (and the carried cache slots rewritten to match) it
    gathers the NEW ids' rows into the same positions -- hits from the cache, misses from the table -- and nothing else changes."""
    from legion_amd import engine, synth
    wl = Workload(scale=12, edge_factor=8, dim=24, n_seeds=900)
    fanout, batch, group = [6, 3], 64, 5
    gpu, cpu = GpuSide(wl, batch, fanout), CpuSide(wl, batch, fanout)
    steps = (wl.sets[(0, 0)][0].size - 1) // batch
    for it in range(steps):
        gpu.run(0, it, 0, is_presc=True); cpu.run(0, it, 0, is_presc=True)
    gpu.cache.candidate_selection(0, gpu.graph)
    gpu.cache.set_capacity(400, 100)
    gpu.cache.fill_up(gpu.feature, gpu.graph)
    cpu.build_cache(0, capacity=(400, 100))
    assert gpu.graph.column_slots(0) == col_slots
    pipe = engine.Pipeline(gpu.graph, gpu.feature, gpu.cache, 0, batch, fanout, group, gpu.pools[0].num_ids, True, 2, weave=True)
    sl = pipe.submit(0, 0)
    pipe.wait(sl)
    before = [engine.read_batch(pipe.pools[sl][lane]) for lane in range(group)]
    ms = pipe.regather_last(sl, 3)
    assert len(ms) == 3 and all(m > 0 for m in ms)
    node_map = gpu.cache.array("node_map", 0)
    H = len(fanout)
    new_ids = []
    for lane in range(group):
        compare_batches(engine.read_batch(pipe.pools[sl][lane]), before[lane], f"regather, lane {lane}: ")
        compare_batches(before[lane], cpu.run(0, lane, 0), f"lane {lane}: ")
        pl = pipe.pools[sl][lane]
        nc = before[lane]["node_counter"]
        a, n = int(nc[9 + H - 1]), int(nc[9 + H] - nc[9 + H - 1])
        ids = torch.from_numpy(np.random.RandomState(lane).permutation(wl.N)[:n].astype(np.int32)).cuda()
        pl.buffer("sampled_ids")[a:a + n] = ids
        if col_slots:
            pl.buffer("node_slot")[a:a + n] = node_map[ids.long()]
        new_ids.append((a, n, ids.cpu().numpy()))
    pipe.regather_last(sl, 1)
    hits = misses = 0
    for lane in range(group):
        a, n, ids = new_ids[lane]
        got = engine.read_batch(pipe.pools[sl][lane])
        assert np.array_equal(got["float_features"][a:a + n].view(np.uint32), wl.features[ids].view(np.uint32))      # the new ids' rows
        assert np.array_equal(got["float_features"][:a].view(np.uint32), before[lane]["float_features"][:a].view(np.uint32))   # earlier hops' rows untouched
        assert np.array_equal(got["agg_src_off"], before[lane]["agg_src_off"]) and np.array_equal(got["node_counter"], before[lane]["node_counter"])
        slots = node_map[torch.from_numpy(ids).long().cuda()].cpu().numpy()
        assert np.array_equal(pipe.pools[sl][lane].buffer("cache_search_buffer")[:n].cpu().numpy(), slots)
        hits += int((slots >= 0).sum()); misses += int((slots < 0).sum())
    assert hits > 0 and misses > 0
    pipe.close()
    gpu.close(); cpu.close()


def test_lane_group_eager(hip, buckets):
    """legion_enqueue_group on caller-owned pools (no pipeline, no graph)."""
    from legion_amd import engine
    wl = Workload(scale=10, edge_factor=8, dim=16, n_seeds=400)
    fanout, batch = [4, 4, 2], 32
    gpu, cpu = GpuSide(wl, batch, fanout), CpuSide(wl, batch, fanout)
    pools = [engine.MemoryPool(0, wl.N, batch, fanout, wl.D) for _ in range(5)]
    for p in pools:
        p.alloc_features(p.num_ids)
    grp = engine.LaneGroup(pools)
    for c0 in (0, 5):
        grp.enqueue(None, gpu.graph, gpu.feature, gpu.cache, batch, c0, 0, 0, fanout)
        torch.cuda.synchronize()
        for lane, p in enumerate(pools):
            compare_batches(engine.read_batch(p), cpu.run(0, c0 + lane, 0), f"lane {lane} of group at {c0}: ")
    grp.close()
    for p in pools:
        p.close()
    gpu.close(); cpu.close()


@pytest.mark.parametrize("P,mode_bits,capacity,replica_rows", [(4, 2, (64, 33), 40), (8, 3, (40, 20), 100), (2, 1, (150, 90), 10_000)])
def test_hot_row_replica_keeps_results_and_saves_peer_reads(hip, P, mode_bits, capacity, replica_rows):
    """Striped clique + a local replica of the clique's hottest rows on every member: batches, hit masks and global slots
    stay bit-identical to the oracle's striped clique (the replica only changes WHERE a hit row is read), and the rows of
    rank < replica size are no longer read through a stripe pointer."""
    wl = Workload(scale=11, edge_factor=8, dim=32, partition_count=P, n_seeds=1200)
    fanout, batch = [5, 4], 64
    gpu, cpu = GpuSide(wl, batch, fanout), CpuSide(wl, batch, fanout)
    steps = min((wl.sets[(p, 0)][0].size - 1) // batch for p in range(P))
    for p in range(P):
        for it in range(steps):
            gpu.run(p, it, 0, is_presc=True); cpu.run(p, it, 0, is_presc=True)
    gpu.cache.candidate_selection(mode_bits, gpu.graph)
    gpu.cache.set_capacity(*capacity)
    gpu.cache.set_replica_memory(replica_rows * wl.D * 4)
    gpu.cache.fill_up(gpu.feature, gpu.graph)
    cpu.build_cache(mode_bits, capacity=capacity)
    Kg = cpu.Kg
    want_rows = min(replica_rows, capacity[0] * Kg)
    for p in range(P):
        assert gpu.cache.replica_rows(p) == want_rows
        assert gpu.cache.gather_stats(p) == (0, 0)                 # enables the counters
    for p in range(P):
        hits = from_replica = 0
        for it in range(2):
            g, c = gpu.run(p, it, 0), cpu.run(p, it, 0)
            compare_batches(g, c, f"replica gpu {p} batch {it}: ")
            assert np.array_equal(g["cache_search_buffer"], c["cache_search_buffer"])
        # hits and their hotness ranks over everything this GPU gathered (all hops of both batches): replay on the host
        node_map = gpu.cache.array("node_map", p).cpu().numpy()
        for it in range(2):
            ids = cpu.run(p, it, 0)["sampled_ids"]
            gslot = node_map[ids]
            hit = gslot >= 0
            rank = (gslot[hit] % capacity[0]) * Kg + gslot[hit] // capacity[0]
            hits += int(hit.sum()); from_replica += int((rank < want_rows).sum())
        stripe, replica = gpu.cache.gather_stats(p)
        assert (stripe, replica) == (hits - from_replica, from_replica) and replica > 0
        # the computed stand-in for the xGMI counter (CostModel's counters[1], SS/engine/server.cu:105-106): of the rows read
        # through a stripe pointer, the ones whose owner g / cap is ANOTHER member of the clique, as 64-byte transactions
        from_peer = 0
        for it in range(2):
            gslot = node_map[cpu.run(p, it, 0)["sampled_ids"]]
            hit = gslot[gslot >= 0]
            rank = (hit % capacity[0]) * Kg + hit // capacity[0]
            from_peer += int(((hit // capacity[0] != p % Kg) & (rank >= want_rows)).sum())
        assert gpu.cache.gather_stats3(p) == (stripe, replica, from_peer) and from_peer <= stripe
        assert from_peer > 0 or want_rows >= capacity[0] * Kg          # (everything replicated: nothing left to read from a peer)
        assert gpu.cache.peer_transactions(p) == from_peer * wl.D * 4 // 64
    gpu.close(); cpu.close()


def test_lds_small_class_picks_16_buckets_on_a_dense_graph(hip, monkeypatch):
    """The small class of the LDS form has 8 or 16 hash buckets per lane; slots say how large a hop CAN get, PreSC says how many
    of them held an edge.  A pipeline created after PreSC on a DENSE graph (every slot of the last hop valid: more claims than 8
    buckets take in one pass) gets 16 buckets, the same shape on a sparse graph 8 -- and both serve what the oracle serves."""
    from legion_amd import engine
    from oracle import ffi
    monkeypatch.delenv("LEGION_LDS_SMALL_BUCKETS", raising=False)
    fanout, batch = [25, 10], 512                                   # hop 2: up to 128 k slots per lane
    for scale, ef, want in ((14, 64, 16), (16, 2, 8)):              # mean degree 64: nearly every slot valid; 2: few are
        wl = Workload(scale=scale, edge_factor=ef, dim=4, n_seeds=4 * batch)
        gpu, cpu = GpuSide(wl, batch, fanout), CpuSide(wl, batch, fanout)
        assert gpu.pools[0].lds_buckets() == 8                      # created before PreSC: nothing known yet
        for it in range(2):                                         # PreSC: the controller records the last hop's maxima
            g, c = gpu.run(0, it, 0, is_presc=True), cpu.run(0, it, 0, is_presc=True)
            compare_batches(g, c, f"presc {it}: ")
        last_edges, before = int(g["edge_counter"][11] - g["edge_counter"][10]), int(g["node_counter"][10])
        assert ((last_edges + before) * 11 // 10 // 8 > 7168) == (want == 16), (last_edges, before)    # (storage.hip's rule)
        gpu.cache.candidate_selection(0, gpu.graph)
        gpu.cache.set_capacity(64, 8)
        gpu.cache.fill_up(gpu.feature, gpu.graph)
        cpu.build_cache(0, capacity=(64, 8))
        pipe = engine.Pipeline(gpu.graph, gpu.feature, gpu.cache, 0, batch, fanout, 2, ffi.num_ids_for(batch, fanout), True, 2)
        assert pipe.pools[0][0].lds_buckets() == want
        slot = pipe.submit(0)
        pipe.wait(slot)
        for lane in range(2):
            compare_batches(engine.read_batch(pipe.pools[slot][lane]), cpu.run(0, lane, 0), f"scale {scale} lane {lane}: ")
        pipe.close(); gpu.close(); cpu.close()


def test_lds_dedup_multi_pass_buckets(hip, monkeypatch):
    """The LDS form when a bucket's vertices do not fit its table: batches of up to ~100 k claims per lane make every
    (lane, bucket) workgroup run 2-4 passes over sub-buckets; still bit-exact, no error raised.  Also a graph with hubs
    sampled thousands of times in one batch (every duplicate lands in the same bucket)."""
    wl = Workload(scale=15, edge_factor=16, dim=4, n_seeds=9000)
    fanout, batch = [10, 10], 2000                         # hop 2: up to 200 k slots per lane
    gpu, cpu = GpuSide(wl, batch, fanout), CpuSide(wl, batch, fanout)
    for it in range(3):
        g, c = gpu.run(0, it, 0), cpu.run(0, it, 0)
        compare_batches(g, c, f"multi-pass lds batch {it}: ")
        assert int(g["edge_counter"][11] - g["edge_counter"][10]) > 8 * 7168      # more claims than 8 tables filled to 14/16 hold
    assert gpu.pools[0].error() == 0
    gpu.close(); cpu.close()
    # a star-heavy graph: vertex 0 is everybody's neighbour many times over
    N = 4096
    deg = np.full(N, 24, dtype=np.int64)
    indptr = np.zeros(N + 1, dtype=np.int64); np.cumsum(deg, out=indptr[1:])
    rng = np.random.RandomState(9)
    col = rng.randint(0, N, indptr[-1]).astype(np.int32)
    col[rng.rand(col.size) < 0.5] = 0
    wl = Workload(dim=4, n_seeds=2000, indptr=indptr, col=col)
    gpu, cpu = GpuSide(wl, 512, [12, 6]), CpuSide(wl, 512, [12, 6])
    for it in range(3):
        compare_batches(gpu.run(0, it, 0), cpu.run(0, it, 0), f"hub graph lds batch {it}: ")
    assert gpu.pools[0].error() == 0
    gpu.close(); cpu.close()


def test_lds_dedup_skewed_sub_bucket(hip, monkeypatch):
    """The LDS form when the hash does NOT spread a bucket over its sub-buckets: every neighbour of every seed hashes to the
    same low 10 bits, so the passes the kernel plans from the claim count (sub-buckets by hash bits 3, 4) all land in one
    sub-bucket of ~11 k distinct vertices -- more than the 8192-word table.  The kernel notices the failed insert and redoes
    the bucket with more passes until the sub-buckets fit: bit-exact, no error (this used to raise
    LG_ERR_TABLE_FULL and leave garbage positions).  The hops here have <= 61 440 slots: this is the retry path of
    dedup_lists_kernel (the barrier behind `passes <<= 1`) -- one template for all four bucket classes since round 5."""
    N = 1 << 24
    x = np.arange(N, dtype=np.uint32)
    x ^= x >> np.uint32(16); x *= np.uint32(0x7feb352d); x ^= x >> np.uint32(15); x *= np.uint32(0x846ca68b); x ^= x >> np.uint32(16)
    S = np.nonzero((x & np.uint32(0x3FF)) == 5)[0].astype(np.int32)              # ~16 k vertices, one bucket, one sub-bucket
    assert S.size > 14_000
    seeds = np.setdiff1d(np.arange(6000, dtype=np.int32), S)[:4096]
    deg = np.zeros(N, dtype=np.int64)
    deg[seeds] = 24
    deg[S] = 24
    indptr = np.zeros(N + 1, dtype=np.int64); np.cumsum(deg, out=indptr[1:])
    col = S[np.random.RandomState(3).randint(0, S.size, int(indptr[-1]))].astype(np.int32)
    wl = Workload(dim=4, n_seeds=16, n_valid=0, n_test=0, indptr=indptr, col=col)
    wl.sets[(0, 0)] = (np.ascontiguousarray(seeds), np.ascontiguousarray(wl.labels_all[seeds]))
    fanout, batch = [20, 3], 1024
    gpu, cpu = GpuSide(wl, batch, fanout), CpuSide(wl, batch, fanout)
    for it in range(2):
        g, c = gpu.run(0, it, 0), cpu.run(0, it, 0)
        compare_batches(g, c, f"skewed sub-bucket batch {it}: ")
        assert int(g["node_counter"][10] - g["node_counter"][9]) > 8192          # hop 1 alone adds more vertices than a table holds
    assert gpu.pools[0].error() == 0
    gpu.close(); cpu.close()


@pytest.mark.parametrize("known_cap", [None, "1", "80"], ids=["lists", "no-room", "some-buckets-overflow"])
def test_lds_dedup_known_lists(hip, monkeypatch, known_cap):
    """LDS form, three hops: hops 2 and 3 recognise the nodes hops 1 and 2 added through the per-bucket lists scatter
    appends to.  A list that outgrows its capacity is not used (that bucket's workgroup scans sampled_ids instead):
    forced for every bucket (capacity 1) and for some of them (capacity 80 against ~75 nodes per bucket)."""
    if known_cap is not None:
        monkeypatch.setenv("LEGION_LDS_KNOWN_CAP", known_cap)
    wl = Workload(scale=12, edge_factor=8, dim=4, n_seeds=600)
    fanout, batch = [4, 3, 3], 48
    gpu, cpu = GpuSide(wl, batch, fanout), CpuSide(wl, batch, fanout)
    for it in range(6):
        compare_batches(gpu.run(0, it, 0), cpu.run(0, it, 0), f"known lists ({known_cap}) batch {it}: ")
    assert gpu.pools[0].error() == 0
    gpu.close(); cpu.close()


@pytest.mark.parametrize("buckets", ["8", "16"])
@pytest.mark.parametrize("claim_cap", ["1", "40", "700"], ids=["no-room", "some-buckets-overflow", "passes"])
def test_lds_dedup_claim_lists_overflow(hip, monkeypatch, claim_cap, buckets):
    """LDS form, 8/16-bucket classes: the sampling kernel appends a hop's claims to one list per hash bucket.  A list that
    cannot take all of its bucket's claims says so by its count, and that bucket's workgroup reads the hop's slots instead:
    forced for every bucket (capacity 1), for some of them (capacity 40 against ~45 claims per bucket of hop 2), and -- on a
    batch whose buckets need several passes over the table -- with the slots walked once per sweep and pass."""
    monkeypatch.setenv("LEGION_LDS_CLAIM_CAP", claim_cap)
    monkeypatch.setenv("LEGION_LDS_SMALL_BUCKETS", buckets)
    if claim_cap == "700":
        wl = Workload(scale=16, edge_factor=16, dim=4, n_seeds=5000)
        fanout, batch, n_it = [12, 12], 1500, 2                # ~150 k claims in hop 2: ~19 k per bucket of 8 -> 4 passes
    else:
        wl = Workload(scale=12, edge_factor=8, dim=4, n_seeds=600)
        fanout, batch, n_it = [4, 3, 3], 48, 6
    gpu, cpu = GpuSide(wl, batch, fanout), CpuSide(wl, batch, fanout)
    for it in range(n_it):
        compare_batches(gpu.run(0, it, 0), cpu.run(0, it, 0), f"claim lists (cap {claim_cap}, {buckets} buckets) batch {it}: ")
    assert gpu.pools[0].error() == 0
    gpu.close(); cpu.close()


@pytest.mark.parametrize("part_wg", [None, "1", "100000000"], ids=["default-tiles", "largest-tiles", "smallest-tiles"])
@pytest.mark.parametrize("batch,fanout", [(6000, [10, 10]), (5000, [5, 5, 5]), (6000, [10, 10, 8]), (8000, [13, 13, 13])],
                         ids=["b6000-10x10-64buckets", "b5000-5x5x5-64buckets", "b6000-10x10x8-256buckets", "b8000-13x13x13-17Mslots"])
def test_lds_dedup_large_batches(hip, monkeypatch, batch, fanout, part_wg):
    """Hops of more than 2^19 slots per lane (Legion's default B = 8000 class): 64 buckets per lane up to 2^22 slots, 256 beyond
    (here 4.8 M and 17.6 M slots).  64 buckets: the sampling kernel stages a super tile's claims in LDS grouped by bucket and
    writes them to the buckets' lists itself.  256 buckets: it samples partition tiles of 4-8 super tiles and reserves, per bucket, a
    run of the bucket's claim list; place_kernel stages the tile's pairs in LDS and writes the runs.  Bit-exact like the 8-bucket class."""
    if part_wg is not None:                              # partition tiles of 8 super tiles, or as few as the class allows
        monkeypatch.setenv("LEGION_LDS_PART_WG", part_wg)
    wl = Workload(scale=16, edge_factor=16, dim=4, n_seeds=3 * batch + 17)
    gpu, cpu = GpuSide(wl, batch, fanout), CpuSide(wl, batch, fanout)
    assert gpu.pools[0].lds_buckets() in (64, 256) and gpu.pools[0].state_bytes() > (1 << 19) * 8
    for it in range(4):                                  # the last batch is the clamped one
        compare_batches(gpu.run(0, it, 0), cpu.run(0, it, 0), f"large lds batch {it}: ")
    assert gpu.pools[0].error() == 0
    gpu.close(); cpu.close()


@pytest.mark.parametrize("batch,fanout,claim_cap,known_cap", [(6000, [10, 10], "1", None), (6000, [10, 10], "2500", None),
                                                              (5000, [5, 5, 5], "1500", "1"), (6000, [10, 10, 8], "1", None),
                                                              (6000, [10, 10, 8], "6000", "900")],
                         ids=["64buckets-no-room", "64buckets-some-overflow", "64buckets-3hops-known-scan", "256buckets-no-room",
                              "256buckets-some-overflow"])
def test_lds_dedup_large_batches_list_overflow(hip, monkeypatch, batch, fanout, claim_cap, known_cap):
    """64- / 256-bucket classes (round 5: one claim list per bucket here too, written by the sampling kernel -- 64 buckets -- or by
    place_kernel into the runs the sampling kernel reserved -- 256): a list that cannot take all of its bucket's claims says so by its count and that bucket's workgroup reads
    the hop's slots instead -- forced for every bucket (capacity 1) and for some (a capacity near the buckets' mean); a known
    list that outgrew its capacity makes its bucket's workgroup scan sampled_ids."""
    monkeypatch.setenv("LEGION_LDS_CLAIM_CAP", claim_cap)
    if known_cap is not None:
        monkeypatch.setenv("LEGION_LDS_KNOWN_CAP", known_cap)
    wl = Workload(scale=16, edge_factor=16, dim=4, n_seeds=2 * batch + 17)
    gpu, cpu = GpuSide(wl, batch, fanout), CpuSide(wl, batch, fanout)
    assert gpu.pools[0].lds_buckets() in (64, 256)
    for it in range(3):                                  # the last batch is the clamped one
        compare_batches(gpu.run(0, it, 0), cpu.run(0, it, 0), f"large batch, list caps {claim_cap} / {known_cap}, batch {it}: ")
    assert gpu.pools[0].error() == 0
    gpu.close(); cpu.close()


@pytest.mark.parametrize("batch,scale,per_thread,buckets", [(6000, 16, 10, 64), (9000, 16, 20, 64), (18000, 17, 10, 256)])
def test_lds_dedup_big_buckets_after_presc(hip, batch, scale, per_thread, buckets):
    """A pipeline created AFTER PreSC knows how many claims the largest hop really has: 0.45 M (0.67 M) in a hop of 1.4 M (2.2 M)
    slots -- more than 5 x 1024 per bucket of 64 -- so its lanes get 64 buckets and the last hop's de-duplication keeps 10 (20)
    claims per thread in registers: dedup_lists_kernel<6,10,13> (64 KB table, two workgroups per CU) for buckets of up to 10 k
    claims, dedup_lists_kernel<6,20,14> (128 KB table, passes over sub-buckets from the registers) beyond.  Third case: more than
    64 x 20 x 1024 claims -- 256 buckets (place_kernel) and, their buckets holding more than 5 k claims, dedup_lists_kernel<8,10,13>.
    Every lane of a group against the oracle."""
    from legion_amd import engine
    fanout, group = [10, 6, 4], 3 if batch < 10000 else 2
    wl = Workload(scale=scale, edge_factor=16, dim=4, n_seeds=(group + 1) * batch + 17)
    gpu, cpu = GpuSide(wl, batch, fanout), CpuSide(wl, batch, fanout)
    last_edges = 0
    for it in range(2):
        g, c = gpu.run(0, it, 0, is_presc=True), cpu.run(0, it, 0, is_presc=True)
        compare_batches(g, c, f"presc {it}: ")
        last_edges = max(last_edges, int(g["edge_counter"][12] - g["edge_counter"][11]))
    lo, hi = {(10, 64): (5, 10), (20, 64): (10, 20), (10, 256): (20, 160)}[(per_thread, buckets)]
    assert 64 * lo * 1024 * 10 // 11 < last_edges <= 64 * hi * 1024 * 10 // 11, last_edges        # (operators.hip / storage.hip's rules)
    gpu.cache.candidate_selection(0, gpu.graph)
    gpu.cache.set_capacity(2000, 200)
    gpu.cache.fill_up(gpu.feature, gpu.graph)
    cpu.build_cache(0, capacity=(2000, 200))
    pipe = engine.Pipeline(gpu.graph, gpu.feature, gpu.cache, 0, batch, fanout, group, gpu.pools[0].num_ids, True, 1, weave=True)
    assert pipe.pools[0][0].lds_buckets() == buckets
    slot = pipe.submit(0, 0)
    pipe.wait(slot)
    for lane in range(group):
        compare_batches(engine.read_batch(pipe.pools[slot][lane]), cpu.run(0, lane, 0), f"big buckets lane {lane}: ")
        assert pipe.pools[slot][lane].error() == 0
    pipe.close()
    gpu.close(); cpu.close()


def test_graph_cache_keeps_modes_and_lane_counts_apart(hip):
    """the hipGraph cache of a pipeline slot used to pack (mode << 40 | active lanes << 32 | batch) into one
    word -- written when groups had at most 128 lanes.  With up to 512 lanes, (mode 0, 256 + 3 lanes) and (mode 1, 3 lanes) of
    one batch size met on the same key: the graph captured for the training group was replayed for the validation group (wrong
    seed set, wrong grid).  Both groups, same slot, same batch size: each must serve what the oracle serves."""
    from legion_amd import engine
    wl = Workload(scale=11, edge_factor=8, dim=4, n_seeds=1700, n_valid=200)
    fanout, batch, G = [3, 2], 6, 260
    gpu, cpu = GpuSide(wl, batch, fanout), CpuSide(wl, batch, fanout)
    pipe = engine.Pipeline(gpu.graph, gpu.feature, gpu.cache, 0, batch, fanout, G, gpu.pools[0].num_ids, True, 1)
    slot = pipe.submit(0, 0, 259)                        # captures (train, 259 lanes)
    pipe.wait(slot)
    for lane in (0, 3, 258):
        compare_batches(engine.read_batch(pipe.pools[slot][lane]), cpu.run(0, lane, 0), f"train group lane {lane}: ")
    slot2 = pipe.submit(0, 1, 3)                         # (valid, 3 lanes): a graph of its own
    assert slot2 == slot
    pipe.wait(slot2)
    for lane in range(3):
        compare_batches(engine.read_batch(pipe.pools[slot][lane]), cpu.run(0, lane, 1), f"validation group lane {lane}: ")
    pipe.close()
    gpu.close(); cpu.close()


def test_pipeline_partial_group(hip):
    """run_range with a length that is not a multiple of the group size: the tail group runs with
    fewer active lanes (its own graph)."""
    from legion_amd import engine
    wl = Workload(scale=10, edge_factor=8, dim=8, n_seeds=600)
    fanout, batch = [4, 3], 32
    gpu, cpu = GpuSide(wl, batch, fanout), CpuSide(wl, batch, fanout)
    pipe = engine.Pipeline(gpu.graph, gpu.feature, gpu.cache, 0, batch, fanout, 4, gpu.pools[0].num_ids, True, 2)
    for first, count in ((0, 7), (3, 9), (0, 2)):
        pipe.run_range(first, count)
        pipe.wait()
        last_group_first = first + (count - 1) // 4 * 4
        slot = ((count + 3) // 4 - 1 + getattr(test_pipeline_partial_group, "_subs", 0)) % 2
        test_pipeline_partial_group._subs = getattr(test_pipeline_partial_group, "_subs", 0) + (count + 3) // 4
        for lane in range(count - (last_group_first - first)):
            compare_batches(engine.read_batch(pipe.pools[slot][lane]), cpu.run(0, last_group_first + lane, 0),
                            f"range {first}+{count} tail lane {lane}: ")
    pipe.close()
    gpu.close(); cpu.close()


@pytest.mark.parametrize("P,mode_bits,capacity,replica_rows,D", [(2, 1, (150, 90), 0, 32), (4, 2, (64, 33), 0, 100), (8, 3, (40, 20), 0, 24),
                                                                 (4, 2, (64, 33), 40, 32), (4, 1, (100, 50), 0, 7)])
def test_owner_bucketed_bulk_transfer_matches_direct_peer_loads(hip, monkeypatch, col_slots, P, mode_bits, capacity, replica_rows, D):
    """LegionTuning.peer_gather = bulk (pipeline.hip, kernels_gather.hip): every member lists, per owner, the rows of that owner's
    stripe its launch group needs; the owners read their own HBM and push whole rows into the requesters' lane arenas.  Batches,
    hit masks, global slots and rows are those of the oracle's striped clique (= what direct peer loads give), on cliques of
    2 / 4 / 8 logical GPUs (two cliques of two in the last case), with and without a hot-row replica, widths 7 / 24 / 32 / 100."""
    from legion_amd import engine
    wl = Workload(scale=11, edge_factor=8, dim=D, partition_count=P, n_seeds=1600)
    fanout, batch, G = [5, 4], 48, 3
    gpu, cpu = GpuSide(wl, batch, fanout), CpuSide(wl, batch, fanout)
    steps = min((wl.sets[(p, 0)][0].size - 1) // batch for p in range(P))
    for p in range(P):
        for it in range(steps):
            gpu.run(p, it, 0, is_presc=True); cpu.run(p, it, 0, is_presc=True)
    gpu.cache.candidate_selection(mode_bits, gpu.graph)
    gpu.cache.set_capacity(*capacity)
    if replica_rows:
        gpu.cache.set_replica_memory(replica_rows * D * 4)
    gpu.cache.fill_up(gpu.feature, gpu.graph)
    cpu.build_cache(mode_bits, capacity=capacity)
    Kg = cpu.Kg
    from oracle import ffi
    rows = ffi.num_ids_for(batch, fanout)
    if D == 100:                                         # (one case with plain arenas; the others: shuffled chunks every member is granted access to)
        monkeypatch.setenv("LEGION_ARENA_SCATTER_MB", "0")
    pipes = [engine.Pipeline(gpu.graph, gpu.feature, gpu.cache, p, batch, fanout, G, rows, use_graph=False, slots=2, arena="shared")
             for p in range(P)]
    for pl in pipes:
        pl.bulk_enable()
    for p in range(P):                                   # every member knows the other members of ITS clique
        for q in range(p // Kg * Kg, p // Kg * Kg + Kg):
            if q != p:
                pipes[p].bulk_link(pipes[q])
    n_groups = min(2, (steps - 1) // G)
    assert n_groups >= 1
    listed = 0
    for grp in range(n_groups):
        slots = [pipes[p].bulk_phase_a(grp * G) for p in range(P)]          # (phase A synchronises its stream: the lists are final)
        assert len(set(slots)) == 1
        listed += sum(pipes[p].bulk_listed(slots[0]) for p in range(P))
        for p in range(P):
            pipes[p].bulk_phase_b(slots[0])                                  # every member pushes what the others listed for it
        for p in range(P):
            for lane in range(G):
                g = engine.read_batch(pipes[p].pools[slots[0]][lane])
                c = cpu.run(p, grp * G + lane, 0)
                compare_batches(g, c, f"bulk gpu {p} group {grp} lane {lane}: ")
                got = pipes[p].pools[slots[0]][lane].buffer("cache_search_buffer")[:max(int(g["node_counter"][1]), 0)].cpu().numpy()
                assert np.array_equal(got, c["cache_search_buffer"])
    assert listed > 0                                    # rows did travel by the bulk path
    for pl in pipes:
        pl.close()
    gpu.close(); cpu.close()
