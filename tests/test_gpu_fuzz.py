"""Seeded random shapes through the HIP path and the oracle side by side: odd vertex counts, rows of degree 0, hubs,
duplicate-heavy adjacency, 1-4 hops (1-6 in the hybrid-tier cases), fan-outs 1..12 (also above every degree), batch sizes that do not divide the seed
set (clamped and empty last batches), feature widths that are not multiples of four, every mode -- once per form of the
first-touch state.  Bit-exact or the test names the seed that failed."""
import numpy as np
import pytest

from tests.gpu_harness import CpuSide, GpuSide
from tests.helpers import Workload, compare_batches

pytestmark = pytest.mark.gpu


def random_case(seed, max_hops=4):
    rng = np.random.RandomState(1000 + seed)
    N = int(rng.choice([37, 64, 257, 1000, 4099]))
    kind = seed % 4
    if kind == 0:                                  # many isolated rows, a few hubs
        deg = np.where(rng.rand(N) < 0.5, 0, rng.randint(1, 6, N))
        deg[rng.choice(N, 3, replace=False)] = rng.randint(N // 2, 2 * N)
    elif kind == 1:                                # uniform small degrees
        deg = rng.randint(0, 4, N)
    elif kind == 2:                                # heavy tail
        deg = np.minimum((rng.pareto(1.2, N) * 2).astype(np.int64), 3 * N)
    else:                                          # dense
        deg = rng.randint(8, 40, N)
    deg = deg.astype(np.int64)
    indptr = np.zeros(N + 1, dtype=np.int64)
    np.cumsum(deg, out=indptr[1:])
    col = rng.randint(0, N, int(indptr[-1])).astype(np.int32)
    if kind == 2 and col.size:                     # duplicate-heavy adjacency: half of all entries name ten vertices
        m = rng.rand(col.size) < 0.5
        col[m] = rng.randint(0, 10, int(m.sum()))
    hops = int(rng.randint(1, max_hops + 1))
    fanout = [int(rng.randint(1, 13)) for _ in range(hops)]
    while np.prod(fanout) > 600:                   # keep the oracle's run short
        fanout[int(np.argmax(fanout))] //= 2
    batch = int(rng.choice([1, 7, 33, 100, 256]))
    dim = int(rng.choice([1, 3, 8, 33]))
    n_seeds = int(min(N - 2, batch * 2 + rng.randint(0, batch + 1)))
    n_valid = int(min(max(N - n_seeds, 0), rng.randint(0, 40)))
    n_test = int(min(max(N - n_seeds - n_valid, 0), rng.randint(0, 40)))
    return dict(indptr=indptr, col=col, fanout=fanout, batch=batch, dim=dim, n_seeds=max(n_seeds, 1), n_valid=n_valid, n_test=n_test)


@pytest.mark.parametrize("seed", range(40))
def test_random_shapes_match_the_oracle(hip, buckets, seed):
    c = random_case(seed)
    wl = Workload(dim=c["dim"], n_seeds=c["n_seeds"], n_valid=c["n_valid"], n_test=c["n_test"], indptr=c["indptr"], col=c["col"])
    gpu, cpu = GpuSide(wl, c["batch"], c["fanout"]), CpuSide(wl, c["batch"], c["fanout"])
    ctx = f"seed {seed} (N {wl.N}, E {wl.E}, fan-out {c['fanout']}, batch {c['batch']}, D {c['dim']}, {buckets} buckets): "
    for mode in (0, 1, 2):
        n_ids = wl.sets[(0, mode)][0].size
        steps = (n_ids + c["batch"] - 1) // c["batch"] + 1          # one past the end: the empty batch
        for it in range(steps):
            compare_batches(gpu.run(0, it, mode), cpu.run(0, it, mode), ctx + f"mode {mode} batch {it}: ")
    assert gpu.pools[0].error() == 0
    gpu.close(); cpu.close()


@pytest.mark.parametrize("seed", range(12))
def test_random_shapes_through_cache_and_pipeline(hip, buckets, seed):
    """The same random shapes through the whole path: PreSC epoch -> hotness -> cost model (random cache memory and
    counters) -> fills -> lane groups of random size under hipGraph replay in a random stream arrangement; capacities,
    hit masks, cache slots, rows and every batch compared with the oracle."""
    from legion_amd import engine
    c = random_case(100 + seed)
    rng = np.random.RandomState(7000 + seed)
    dim = max(c["dim"], 3)
    wl = Workload(dim=dim, n_seeds=c["n_seeds"], n_valid=c["n_valid"], n_test=c["n_test"], indptr=c["indptr"], col=c["col"])
    batch, fanout = c["batch"], c["fanout"]
    cache_memory = int(rng.choice([20_000, 200_000, 2_000_000]))
    counters = (int(rng.randint(0, 50_000)), int(rng.randint(0, 50_000))) if seed % 2 else (0, 0)
    gpu, cpu = GpuSide(wl, batch, fanout, cache_memory=cache_memory), CpuSide(wl, batch, fanout)
    ctx = f"seed {seed} (N {wl.N}, E {wl.E}, fan-out {fanout}, batch {batch}, D {dim}, cache {cache_memory}, {buckets} buckets): "
    steps = max((wl.sets[(0, 0)][0].size - 1) // batch, 1)
    for it in range(steps):
        compare_batches(gpu.run(0, it, 0, is_presc=True), cpu.run(0, it, 0, is_presc=True), ctx + f"presc {it}: ")
    gpu.cache.candidate_selection(0, gpu.graph)
    gpu.cache.cost_model(gpu.feature, gpu.graph, counters, steps)
    oc = cpu.build_cache(0, cache_memory=cache_memory, train_step=steps, counters=counters)[0]
    assert (gpu.cache.node_capacity(0), gpu.cache.edge_capacity(0)) == (oc.node_capacity, oc.edge_capacity), ctx
    gpu.cache.fill_up(gpu.feature, gpu.graph)
    group, slots = int(rng.randint(1, 6)), int(rng.randint(1, 4))
    arrangement = ["one-stream", "weave", "weave"][int(rng.randint(0, 3))]      # (the draw keeps the later ones of the sequence where they were)
    pipe = engine.Pipeline(gpu.graph, gpu.feature, gpu.cache, 0, batch, fanout, group, gpu.pools[0].num_ids, True, slots,
                           weave=arrangement == "weave")
    for mode in (0, 1):
        n_batches = (wl.sets[(0, mode)][0].size + batch - 1) // batch + 1
        for g0 in range(0, n_batches, group):
            sl = pipe.submit(g0, mode)
            pipe.wait(sl)
            for lane in range(group):
                got, want = engine.read_batch(pipe.pools[sl][lane]), cpu.run(0, g0 + lane, mode)
                compare_batches(got, want, ctx + f"{arrangement} group {group} mode {mode} batch {g0 + lane}: ")
    assert all(p.error() == 0 for row in pipe.pools for p in row)
    pipe.close()
    gpu.close(); cpu.close()


@pytest.mark.parametrize("seed", range(14))
def test_random_shapes_hybrid_tier_and_up_to_six_hops(hip, buckets, seed):
    """Random shapes with one to SIX hops through the hybrid CPU-cache / GPU-cache tier (UnifiedCache::HybridInit, random capacities
    incl. 0 and beyond N) -- or, every third seed, the clique cache -- and lane groups under hipGraph replay: the order, the map and
    every batch against the oracle."""
    from legion_amd import engine
    from oracle import ffi
    c = random_case(300 + seed, max_hops=6)
    rng = np.random.RandomState(9000 + seed)
    dim = max(c["dim"], 3)
    wl = Workload(dim=dim, n_seeds=c["n_seeds"], n_valid=c["n_valid"], n_test=c["n_test"], indptr=c["indptr"], col=c["col"])
    batch, fanout = c["batch"], c["fanout"]
    gpu, cpu = GpuSide(wl, batch, fanout, cache_memory=200_000), CpuSide(wl, batch, fanout)
    ctx = f"seed {seed} (N {wl.N}, E {wl.E}, fan-out {fanout}, batch {batch}, D {dim}, {buckets} buckets): "
    steps = max((wl.sets[(0, 0)][0].size - 1) // batch, 1)
    for it in range(steps):
        compare_batches(gpu.run(0, it, 0, is_presc=True), cpu.run(0, it, 0, is_presc=True), ctx + f"presc {it}: ")
    if seed % 3 == 2:
        gpu.cache.candidate_selection(0, gpu.graph)
        gpu.cache.cost_model(gpu.feature, gpu.graph, (0, 0), steps)
        gpu.cache.fill_up(gpu.feature, gpu.graph)
        oc = cpu.build_cache(0, cache_memory=200_000, train_step=steps)[0]
    else:
        cpu_cap = int(rng.choice([0, 5, wl.N // 7, wl.N // 2, 2 * wl.N]))
        gpu_cap = int(rng.choice([0, 3, wl.N // 5, wl.N // 2, 2 * wl.N]))
        gpu.cache.hybrid_init(gpu.feature, gpu.graph, cpu_cap, gpu_cap)
        oc = ffi.OracleCache(wl.N, wl.D, 1, 0)
        oc.hybrid_init(cpu.node_access[0], wl.features, cpu_cap, gpu_cap)
        cpu.Kg, cpu.caches = 1, [oc]
        ctx += f"hybrid cpu {cpu_cap} gpu {gpu_cap}: "
    assert np.array_equal(gpu.cache.array("QF", 0).cpu().numpy(), oc.arr("QF", np.int32)), ctx
    assert np.array_equal(gpu.cache.array("node_map", 0).cpu().numpy(), oc.arr("node_map", np.int32)), ctx
    group, slots = int(rng.randint(1, 5)), int(rng.randint(1, 4))
    pipe = engine.Pipeline(gpu.graph, gpu.feature, gpu.cache, 0, batch, fanout, group, gpu.pools[0].num_ids, True, slots, weave=bool(seed % 2))
    for mode in (0, 1, 2):
        n_batches = (wl.sets[(0, mode)][0].size + batch - 1) // batch + 1
        for g0 in range(0, n_batches, group):
            sl = pipe.submit(g0, mode)
            pipe.wait(sl)
            for lane in range(group):
                compare_batches(engine.read_batch(pipe.pools[sl][lane]), cpu.run(0, g0 + lane, mode), ctx + f"mode {mode} batch {g0 + lane}: ")
    assert all(p.error() == 0 for row in pipe.pools for p in row)
    pipe.close()
    gpu.close(); cpu.close()
