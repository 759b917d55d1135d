"""Single-kernel parity through the C ABI: the draw, the gather, the synthetic generators."""
import ctypes
import json
import os

import numpy as np
import pytest
import torch

from legion_amd import synth

pytestmark = pytest.mark.gpu
GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "rng_thrust.json")))


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def test_draw_matches_thrust_golden(hip):
    arr = np.array(GOLD["draws"], dtype=np.int64)
    idx = torch.from_numpy(arr[:, 0].astype(np.int32)).cuda()
    deg = torch.from_numpy(arr[:, 1].astype(np.int32)).cuda()
    out = torch.empty_like(idx)
    hip.legion_draw_batch(_stream(), _p(idx), _p(deg), _p(out), idx.numel())
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), arr[:, 2].astype(np.int32))


def test_draw_matches_oracle_bulk(hip, oracle):
    rng = np.random.RandomState(5)
    n = 200000
    idx = np.concatenate([rng.randint(0, 1 << 23, n), rng.randint(0, 2**31 - 2, n // 4)]).astype(np.int32)
    deg = np.concatenate([rng.randint(1, 2000, n), rng.randint(1, 2**31 - 1, n // 4)]).astype(np.int32)
    ti, td = torch.from_numpy(idx).cuda(), torch.from_numpy(deg).cuda()
    out = torch.empty_like(ti)
    hip.legion_draw_batch(_stream(), _p(ti), _p(td), _p(out), ti.numel())
    torch.cuda.synchronize()
    want = np.array([oracle.lgo_draw(int(i), int(d)) for i, d in zip(idx, deg)], dtype=np.int32)
    assert np.array_equal(out.cpu().numpy(), want)


def test_draw_dense_slots_all_small_degrees(hip, oracle):
    # every slot index below 2^16 at every degree 1..40: the regime every fan-out in BASELINE.json lives in
    idx = np.repeat(np.arange(1 << 16, dtype=np.int32), 40)
    deg = np.tile(np.arange(1, 41, dtype=np.int32), 1 << 16)
    ti, td = torch.from_numpy(idx).cuda(), torch.from_numpy(deg).cuda()
    out = torch.empty_like(ti)
    hip.legion_draw_batch(_stream(), _p(ti), _p(td), _p(out), ti.numel())
    torch.cuda.synchronize()
    x = np.array([pow(48271, int(i) + 1, 2147483647) for i in range(1 << 16)], dtype=np.float64)
    want = ((np.repeat(x, 40) - 1.0) / 2147483646.0 * deg.astype(np.float64)).astype(np.int32)
    assert np.array_equal(out.cpu().numpy(), want)


@pytest.mark.parametrize("D", [100, 128, 256, 602, 7, 4, 1, 1024])
def test_gather_rows(hip, D):
    N, cap, Kg = 5000, 700, 2
    rng = np.random.RandomState(D)
    table = synth.features_numpy(0, N, D, 7)
    node_map = np.full(N, -2, dtype=np.int32)
    cached = rng.permutation(N)[:cap * Kg]
    node_map[cached] = (np.arange(cap * Kg) % Kg) * cap + np.arange(cap * Kg) // Kg
    caches = [np.zeros((cap, D), dtype=np.float32) for _ in range(Kg)]
    for t, v in enumerate(cached):
        caches[t % Kg][t // Kg] = table[v] + np.float32(1000.0)        # make hits distinguishable
    ids = rng.randint(0, N, size=3000).astype(np.int32)
    ids[::97] = -1                                                      # skipped rows (id < 0)
    off, cnt = 123, 2500
    t_table = torch.from_numpy(table).cuda()
    t_caches = [torch.from_numpy(c).cuda() for c in caches]
    ptrs = torch.tensor([c.data_ptr() for c in t_caches], dtype=torch.int64).cuda()
    t_map = torch.from_numpy(node_map).cuda()
    t_ids = torch.from_numpy(ids).cuda()
    t_range = torch.tensor([off, cnt], dtype=torch.int32).cuda()
    dst = torch.full((3000, D), -7.0, dtype=torch.float32).cuda()
    cidx = torch.full((3000,), 99, dtype=torch.int32).cuda()
    hip.legion_gather_rows(_stream(), _p(t_table), _p(ptrs), _p(t_map), cap, D, N, _p(t_ids), _p(cidx), _p(t_range),
                           _p(dst), 3000)
    torch.cuda.synchronize()
    want = np.full((3000, D), -7.0, dtype=np.float32)
    want_idx = np.full(3000, 99, dtype=np.int32)
    for r in range(cnt):
        v = ids[off + r]
        g = node_map[v] if v >= 0 else -2
        want_idx[r] = g
        if g < 0:
            if v >= 0:
                want[off + r] = table[v]
        else:
            want[off + r] = caches[g // cap][g % cap]
    assert np.array_equal(cidx.cpu().numpy(), want_idx)
    assert np.array_equal(dst.cpu().numpy().view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("rows_per_wg,D,cnt", [(16, 4, 300_017), (64, 7, 1_200_003), (32, 128, 530_001)])
def test_gather_rows_workgroups_walk_several_tiles(hip, monkeypatch, rows_per_wg, D, cnt):
    """More tiles than workgroups (the grid is capped at ~8192 workgroups per launch): a workgroup then walks tiles
    blockIdx.x, + gridDim.x, ... with the NEXT tile's ids and slot lookups prefetched while it copies (kernels_gather.hip) --
    two and three tiles per workgroup, a partial last tile, hits / misses / skipped rows mixed."""
    monkeypatch.setenv("LEGION_GATHER_ROWS", str(rows_per_wg))
    hip.legion_tuning_from_env()
    try:
        N, cap = 40_000, 9_000
        rng = np.random.RandomState(cnt % 1000)
        table = synth.features_numpy(0, N, D, 7)
        node_map = np.full(N, -2, dtype=np.int32)
        cached = rng.permutation(N)[:cap]
        node_map[cached] = np.arange(cap, dtype=np.int32)
        cache = table[cached] + np.float32(1000.0)                      # make hits distinguishable
        ids = rng.randint(0, N, size=cnt).astype(np.int32)
        ids[::101] = -1                                                  # skipped rows
        t_table, t_cache = torch.from_numpy(table).cuda(), torch.from_numpy(cache).cuda()
        ptrs = torch.tensor([t_cache.data_ptr()], dtype=torch.int64).cuda()
        t_map, t_ids = torch.from_numpy(node_map).cuda(), torch.from_numpy(ids).cuda()
        t_range = torch.tensor([0, cnt], dtype=torch.int32).cuda()
        dst = torch.full((cnt, D), -7.0, dtype=torch.float32).cuda()
        cidx = torch.full((cnt,), 99, dtype=torch.int32).cuda()
        hip.legion_gather_rows(_stream(), _p(t_table), _p(ptrs), _p(t_map), cap, D, N, _p(t_ids), _p(cidx), _p(t_range), _p(dst), cnt)
        torch.cuda.synchronize()
        g = np.where(ids >= 0, node_map[np.maximum(ids, 0)], -2).astype(np.int32)
        want = np.where((g >= 0)[:, None], cache[np.maximum(g, 0)], table[np.maximum(ids, 0)])
        want[ids < 0] = -7.0
        assert np.array_equal(cidx.cpu().numpy(), g)
        assert np.array_equal(dst.cpu().numpy().view(np.uint32), want.astype(np.float32).view(np.uint32))
    finally:
        monkeypatch.delenv("LEGION_GATHER_ROWS")
        hip.legion_tuning_from_env()


def test_gather_no_map_all_miss_and_zero_rows(hip):
    N, D = 1000, 128
    table = torch.from_numpy(synth.features_numpy(0, N, D, 7)).cuda()
    ids = torch.arange(N - 1, -1, -1, dtype=torch.int32).cuda()
    dst = torch.zeros((N, D)).cuda()
    cidx = torch.zeros(N, dtype=torch.int32).cuda()
    rng_dev = torch.tensor([0, N], dtype=torch.int32).cuda()
    hip.legion_gather_rows(_stream(), _p(table), None, None, 1, D, N, _p(ids), _p(cidx), _p(rng_dev), _p(dst), N)
    torch.cuda.synchronize()
    assert torch.equal(dst, table.flip(0)) and bool((cidx == -2).all())
    rng_dev = torch.tensor([5, 0], dtype=torch.int32).cuda()
    dst.zero_()
    hip.legion_gather_rows(_stream(), _p(table), None, None, 1, D, N, _p(ids), _p(cidx), _p(rng_dev), _p(dst), N)
    torch.cuda.synchronize()
    assert not dst.any()


def test_synth_generators_match_numpy(hip):
    indptr, col = synth.rmat_csr_device(12, 8, 20231)
    ip, cl = synth.rmat_csr_numpy(12, 8, 20231)
    assert np.array_equal(indptr.cpu().numpy(), ip) and np.array_equal(col.cpu().numpy(), cl)
    indptr, col = synth.rmat_csr_device(13, 4, 20231, scramble=True)          # Graph500-style label scrambling
    ip, cl = synth.rmat_csr_numpy(13, 4, 20231, scramble=True)
    assert np.array_equal(indptr.cpu().numpy(), ip) and np.array_equal(col.cpu().numpy(), cl)
    f = synth.features_device(3000, 100, 7)
    assert np.array_equal(f.cpu().numpy().view(np.uint32), synth.features_numpy(0, 3000, 100, 7).view(np.uint32))
    ids = torch.tensor([5, 17, -1, 2999], dtype=torch.int32).cuda()
    rows = f[ids.clamp(min=0).long()].contiguous()
    assert synth.feature_check_device(rows, ids, 100, 7) == 0
    rows[1, 3] += 1.0
    assert synth.feature_check_device(rows, ids, 100, 7) == 1
