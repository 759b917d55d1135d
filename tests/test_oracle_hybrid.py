"""The oracle's hybrid CPU-cache / GPU-cache tier (lgo_hybrid_init, the hybrid branch of lgo_feature_cache_lookup) against an
independent numpy restatement of the same reference lines: SS/cache/cache.cu:614-670 (HybridInit), :138-153 (HybridInsert),
SS/cache/cache_impl.cuh:113-123 (HybridInitPair), :202-235 (feat_cache_lookup).  The reference holds no vectors for this tier
(it never calls it, SS/engine/server.cu:112): like the rest of the hot path, "parity unpinned" against the reference itself."""
import numpy as np
import pytest

from oracle import ffi


def np_hybrid(node_access, features, cpu_cap, gpu_cap):
    """(QF, node_map, gpu_cache, cpu_cache) from one GPU's counters."""
    N, D = features.shape
    order = np.argsort(-node_access.astype(np.int64), kind="stable").astype(np.int32)   # sort_by_key(greater) over iota: ties by id
    node_map = np.full(N, -2, dtype=np.int32)
    t = np.arange(min(cpu_cap + gpu_cap, N))
    node_map[order[t]] = np.where(t < gpu_cap, cpu_cap + t, t - gpu_cap)                 # HybridInitPair
    gpu_cache = np.zeros((max(gpu_cap, 1), D), dtype=np.float32)
    cpu_cache = np.zeros((max(cpu_cap, 1), D), dtype=np.float32)
    n_gpu = min(gpu_cap, N)
    n_cpu = max(min(cpu_cap, N - gpu_cap), 0)
    gpu_cache[:n_gpu] = features[order[:n_gpu]]
    cpu_cache[:n_cpu] = features[order[gpu_cap:gpu_cap + n_cpu]]
    return order, node_map, gpu_cache, cpu_cache


def np_lookup(ids, node_map, gpu_cache, cpu_cache, cpu_cap, gpu_cap, dst, table):
    """feat_cache_lookup row by row; table None = the kernel to the letter (a miss row is not written)."""
    for r, v in enumerate(ids):
        g = -2 if v < 0 else int(node_map[v])
        if 0 <= g < cpu_cap:
            dst[r] = cpu_cache[g % cpu_cap]
        elif g >= cpu_cap:
            dst[r] = gpu_cache[(g - cpu_cap) % gpu_cap]
        elif table is not None and v >= 0:
            dst[r] = table[v % table.shape[0]]


@pytest.mark.parametrize("N,D,cpu_cap,gpu_cap", [(500, 8, 60, 40), (300, 5, 0, 100), (300, 16, 100, 0), (64, 4, 50, 50), (200, 3, 1, 1)])
@pytest.mark.parametrize("with_table", [True, False])
def test_hybrid_init_and_lookup_match_numpy(oracle, N, D, cpu_cap, gpu_cap, with_table):
    rng = np.random.RandomState(N + D + cpu_cap)
    access = (rng.zipf(1.6, N) % 50).astype(np.uint64)
    access[rng.rand(N) < 0.5] = 0                                       # most vertices tie at hotness 0, as after PreSC
    feats = rng.rand(N, D).astype(np.float32)
    c = ffi.OracleCache(N, D, 1, 0)
    c.hybrid_init(access, feats, cpu_cap, gpu_cap)
    order, node_map, gpu_cache, cpu_cache = np_hybrid(access, feats, cpu_cap, gpu_cap)
    assert np.array_equal(c.arr("QF", np.int32), order)
    assert np.array_equal(c.arr("node_map", np.int32), node_map)
    assert np.all(c.arr("edge_index_map", np.int8) == -2) and np.all(c.arr("edge_offset_map", np.int32) == -2)
    cc = c.c.contents
    assert (cc.hybrid, cc.cpu_cache_capacity, cc.gpu_cache_capacity, cc.node_capacity, cc.edge_capacity) == (1, cpu_cap, gpu_cap, cpu_cap + gpu_cap, 0)
    assert np.array_equal(np.ctypeslib.as_array(cc.feat_cache[0], shape=(max(gpu_cap, 1) * D,)).reshape(-1, D), gpu_cache)
    assert np.array_equal(np.ctypeslib.as_array(cc.cpu_cache, shape=(max(cpu_cap, 1) * D,)).reshape(-1, D), cpu_cache)

    # the lookup over one "hop" of ids, including a skipped (negative) id, through lgo_find_feat + lgo_feature_cache_lookup
    ids = np.concatenate([rng.permutation(N)[:min(N, 150)], [-1]]).astype(np.int32)
    pool = ffi.OraclePool(N, ids.size, [1], ids.size * 2, D)
    p = pool.p.contents
    np.ctypeslib.as_array(p.sampled_ids, shape=(ids.size,))[:] = ids
    p.node_counter[0], p.node_counter[1] = 0, ids.size                  # the new-node range of op 0 (operator_impl.cu:60-66)
    out = np.ctypeslib.as_array(p.float_features, shape=(ids.size * 2 * D,)).reshape(-1, D)
    out[:] = -5.0
    L = oracle
    L.lgo_find_feat(c.c, pool.p, 0)
    L.lgo_feature_cache_lookup(c.c, pool.p, ffi._p(feats if with_table else None, ffi.P_F32), 1)
    want = np.full((ids.size * 2, D), -5.0, dtype=np.float32)
    np_lookup(ids, node_map, gpu_cache, cpu_cache, cpu_cap, gpu_cap, want, feats if with_table else None)
    assert np.array_equal(np.ctypeslib.as_array(p.cache_search_buffer, shape=(ids.size,)),
                          np.where(ids >= 0, node_map[np.maximum(ids, 0)], -2))
    assert np.array_equal(out.view(np.uint32), want.view(np.uint32))
    assert (p.node_counter[2], p.node_counter[3]) == (0, ids.size)      # counter_update(op % 3 == 1), operator_impl.cu:83-85
    pool.close(); c.close()
