"""Synthetic workloads: RMAT graphs in the reference's CSR layout
(int64 indptr[N+1], int32 col[E]), counter-hash float32 features, seeded seed sets.

Two generators that produce the SAME values:
  * device: HIP kernels in liblegion_hip.so (legion_synth_*), used at bench sizes (RMAT-26, 34 GB
    of features never touch the host);
  * numpy: a restatement of the same integer hash, used by CPU-only tests and to check the device
    generators.
"""
import ctypes

import numpy as np

RMAT_A, RMAT_AB, RMAT_ABC = int(0.57 * 65536.0), int(0.76 * 65536.0), int(0.95 * 65536.0)
_M64 = (1 << 64) - 1


def splitmix64_np(x):
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return x ^ (x >> np.uint64(31))


SCRAMBLE_KEY = 0x9E3779B97F4A7C15 ^ 20231     # bench.py --scramble / tests: any non-zero 64-bit key


def scramble_labels_numpy(x, scale, key):
    """kernels_synth.hip: scramble_label -- a bijection of [0, 2^scale)."""
    mask = np.uint64((1 << scale) - 1)
    m1 = np.uint64((key & 0xFFFFFFFF) | 1)
    m2 = np.uint64(((key >> 32) & 0xFFFFFFFF) | 1)
    add = np.uint64((key >> 17) & 0xFFFFFFFF)
    sh = np.uint64((scale + 1) // 2 if scale > 1 else 1)
    x = np.asarray(x, dtype=np.uint64)
    x = (x * m1 + add) & mask
    x = x ^ (x >> sh)
    x = (x * m2) & mask
    x = x ^ (x >> sh)
    return (x & mask).astype(np.uint32)


def rmat_edges_numpy(scale, num_edges, seed, scramble_key=0):
    """Same edges as legion_synth_rmat_edges[_scrambled] (kernels_synth.hip: rmat_kernel)."""
    e = np.arange(num_edges, dtype=np.uint64)
    u = np.zeros(num_edges, dtype=np.uint32)
    v = np.zeros(num_edges, dtype=np.uint32)
    h = None
    for level in range(scale):
        if level & 3 == 0:
            with np.errstate(over="ignore"):
                h = splitmix64_np(np.uint64(seed) ^ (e * np.uint64(8) + np.uint64(level >> 2)))
        r = ((h >> np.uint64((level & 3) * 16)) & np.uint64(0xFFFF)).astype(np.uint32)
        ubit = (r >= RMAT_AB).astype(np.uint32)
        vbit = (((r >= RMAT_A) & (r < RMAT_AB)) | (r >= RMAT_ABC)).astype(np.uint32)
        u = (u << np.uint32(1)) | ubit
        v = (v << np.uint32(1)) | vbit
    same = u == v
    v[same] = u[same] ^ np.uint32(1)
    if scramble_key:
        u, v = scramble_labels_numpy(u, scale, scramble_key), scramble_labels_numpy(v, scale, scramble_key)
    return u.astype(np.int32), v.astype(np.int32)


def csr_from_edges_numpy(num_nodes, src, dst):
    order = np.argsort(src, kind="stable")
    col = np.ascontiguousarray(dst[order], dtype=np.int32)
    counts = np.bincount(src, minlength=num_nodes).astype(np.int64)
    indptr = np.zeros(num_nodes + 1, dtype=np.int64)
    np.cumsum(counts, out=indptr[1:])
    return indptr, col


def rmat_csr_numpy(scale, edge_factor, seed, scramble=False):
    n = 1 << scale
    src, dst = rmat_edges_numpy(scale, n * edge_factor, seed, SCRAMBLE_KEY if scramble else 0)
    return csr_from_edges_numpy(n, src, dst)


def features_numpy(first_row, num_rows, dim, seed):
    """Same values as legion_synth_features (kernels_synth.hip: synth_feature_value)."""
    idx = (np.arange(first_row, first_row + num_rows, dtype=np.int64)[:, None] * dim +
           np.arange(dim, dtype=np.int64)[None, :]).astype(np.uint64)
    h = splitmix64_np(np.uint64(seed) ^ idx)
    k = (h >> np.uint64(40)).astype(np.int64)
    return (k.astype(np.float32) * np.float32(1.0 / 8388608.0) - np.float32(1.0)).astype(np.float32)


def feature_rows_numpy(ids, dim, seed):
    ids = np.asarray(ids, dtype=np.int64)
    idx = (ids[:, None] * dim + np.arange(dim, dtype=np.int64)[None, :]).astype(np.uint64)
    h = splitmix64_np(np.uint64(seed) ^ idx)
    k = (h >> np.uint64(40)).astype(np.int64)
    return (k.astype(np.float32) * np.float32(1.0 / 8388608.0) - np.float32(1.0)).astype(np.float32)


def seed_ids(num_nodes, count, seed):
    """First `count` entries of a seeded permutation of [0, num_nodes)."""
    count = min(int(count), int(num_nodes))
    if num_nodes & (num_nodes - 1) == 0:     # power of two: odd-multiplier affine map is a bijection
        a = (int(splitmix64_np(np.uint64(seed))) | 1) % num_nodes
        b = int(splitmix64_np(np.uint64(seed + 1))) % num_nodes
        i = np.arange(count, dtype=np.uint64)
        with np.errstate(over="ignore"):
            return ((i * np.uint64(a) + np.uint64(b)) % np.uint64(num_nodes)).astype(np.int32)
    return np.random.RandomState(seed).permutation(num_nodes)[:count].astype(np.int32)


# ---- device generators -------------------------------------------------------------------------
def rmat_csr_device(scale, edge_factor, seed, device="cuda:0", scramble=False):
    import torch
    from . import lib as _libmod
    lib = _libmod.load()
    n = 1 << scale
    e = n * edge_factor
    src = torch.empty(e, dtype=torch.int32, device=device)
    dst = torch.empty(e, dtype=torch.int32, device=device)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    lib.legion_synth_rmat_edges_scrambled(stream, scale, e, seed, ctypes.c_void_p(src.data_ptr()),
                                          ctypes.c_void_p(dst.data_ptr()), SCRAMBLE_KEY if scramble else 0)
    counts = torch.bincount(src, minlength=n)
    indptr = torch.zeros(n + 1, dtype=torch.int64, device=device)
    torch.cumsum(counts, 0, out=indptr[1:])
    del counts
    _, order = torch.sort(src, stable=True)
    del src
    col = dst[order].contiguous()
    del order, dst
    return indptr, col


def csr_device_large(num_nodes, num_edges, seed, device="cuda:0", chunk_edges=1 << 27):
    """A skewed synthetic graph with ANY vertex count and ANY edge count -- in particular more than 2^32 edges and a vertex
    count that is not a power of two, the sizes of the reference's real data sets (legion_server.py:41-88: uk-union
    N = 133 633 040, E = 5 507 679 822; papers100M N = 111 059 956, E = 1 615 685 872) -- built on the device without ever
    holding an edge list: RMAT edges of scale ceil(log2 N) are generated chunk by chunk (every edge a pure function of (seed,
    chunk, index)), endpoints folded into [0, N) by `% N`, counted in a first pass (-> int64 indptr) and placed in a second
    pass over the same chunks (stable sort inside a chunk, a per-row cursor across chunks: deterministic).
    Returns (int64 indptr[N+1], int32 col[E]) on `device`."""
    import torch
    from . import lib as _libmod
    lib = _libmod.load()
    N, E = int(num_nodes), int(num_edges)
    assert 2 <= N < (1 << 31) and E >= 0
    scale = max(1, int(np.ceil(np.log2(N))))
    n_chunks = (E + chunk_edges - 1) // chunk_edges
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def gen(ci):
        n = min(chunk_edges, E - ci * chunk_edges)
        src = torch.empty(n, dtype=torch.int32, device=device)
        dst = torch.empty(n, dtype=torch.int32, device=device)
        cseed = (int(seed) + 0x9E3779B97F4A7C15 * (ci + 1)) & 0xFFFFFFFFFFFFFFFF
        lib.legion_synth_rmat_edges_scrambled(stream, scale, n, cseed, ctypes.c_void_p(src.data_ptr()),
                                              ctypes.c_void_p(dst.data_ptr()), SCRAMBLE_KEY)
        if N != (1 << scale):
            src.remainder_(N)
            dst.remainder_(N)
            loop = src == dst
            dst[loop] = (dst[loop] + 1) % N
        return src, dst

    counts = torch.zeros(N, dtype=torch.int64, device=device)
    for ci in range(n_chunks):
        src, _ = gen(ci)
        counts += torch.bincount(src, minlength=N)
        del src, _
    indptr = torch.zeros(N + 1, dtype=torch.int64, device=device)
    torch.cumsum(counts, 0, out=indptr[1:])
    del counts
    cursor = indptr[:-1].clone()
    col = torch.empty(E, dtype=torch.int32, device=device)
    for ci in range(n_chunks):
        src, dst = gen(ci)
        s, order = torch.sort(src, stable=True)
        d = dst[order]
        del src, dst, order
        uniq, cnt = torch.unique_consecutive(s, return_counts=True)
        starts = torch.cumsum(cnt, 0) - cnt
        rank = torch.arange(s.numel(), dtype=torch.int64, device=device) - torch.repeat_interleave(starts, cnt)
        uniq = uniq.long()
        pos = torch.repeat_interleave(cursor[uniq], cnt) + rank
        col[pos] = d
        cursor[uniq] += cnt
        del s, d, uniq, cnt, starts, rank, pos
    del cursor
    torch.cuda.empty_cache()
    return indptr, col


def features_device(num_rows, dim, seed, device="cuda:0"):
    import torch
    from . import lib as _libmod
    lib = _libmod.load()
    out = torch.empty((num_rows, dim), dtype=torch.float32, device=device)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    lib.legion_synth_features(stream, ctypes.c_void_p(out.data_ptr()), 0, num_rows, dim, seed)
    return out


def features_device_rows(first_row, num_rows, dim, seed, device="cuda:0"):
    """Rows first_row .. first_row+num_rows-1 of the same table (chunked generation of tables that do not live in HBM)."""
    import torch
    from . import lib as _libmod
    lib = _libmod.load()
    out = torch.empty((num_rows, dim), dtype=torch.float32, device=device)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    lib.legion_synth_features(stream, ctypes.c_void_p(out.data_ptr()), int(first_row), int(num_rows), dim, seed)
    return out


def feature_check_device(rows, ids, dim, seed):
    """Number of float32 words in `rows` that differ bitwise from the generator's value for ids."""
    import torch
    from . import lib as _libmod
    lib = _libmod.load()
    bad = torch.zeros(1, dtype=torch.int64, device=rows.device)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    lib.legion_synth_feature_check(stream, ctypes.c_void_p(rows.data_ptr()), ctypes.c_void_p(ids.data_ptr()),
                                   int(ids.numel()), dim, seed, ctypes.c_void_p(bad.data_ptr()))
    return int(bad.item())
