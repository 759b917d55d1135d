"""Shared builders for the parity tests: small synthetic workloads run through the oracle (CPU)
and through liblegion_hip.so (GPU) with identical inputs."""
import numpy as np

from legion_amd import synth
from oracle import ffi

KEYS_EXACT = ["node_counter", "edge_counter", "sampled_ids", "labels", "agg_src_off", "agg_dst_off",
              "agg_src_ids", "agg_dst_ids"]


class Workload:
    def __init__(self, scale=10, edge_factor=8, dim=16, seed=20231, n_seeds=None, n_valid=300, n_test=200,
                 partition_count=1, indptr=None, col=None, partition=None):
        if indptr is None:
            indptr, col = synth.rmat_csr_numpy(scale, edge_factor, seed)
        self.indptr, self.col = indptr, col
        self.N = int(indptr.size - 1)
        self.E = int(col.size)
        self.D = dim
        self.P = partition_count
        self.features = synth.features_numpy(0, self.N, dim, 7) if dim > 0 else None
        n_seeds = n_seeds if n_seeds is not None else max(self.N // 4, 1)
        perm = np.random.RandomState(11).permutation(self.N).astype(np.int32)
        train = perm[:n_seeds]
        valid = perm[n_seeds:n_seeds + n_valid]
        test = perm[n_seeds + n_valid:n_seeds + n_valid + n_test]
        self.labels_all = (np.arange(self.N, dtype=np.int64) * 2654435761 % 47).astype(np.int32)
        # storage_management.cu:171-203: id % partition_count; with a `partition` file the TRAINING ids go where the
        # file says (an entry >= partition_count drops the id), validation and testing ids stay on id % partition_count
        self.train, self.valid, self.test = train, valid, test
        self.sets = {}
        for mode, ids in ((0, train), (1, valid), (2, test)):
            part_of = partition[ids] if (partition is not None and mode == 0) else ids % partition_count
            for p in range(partition_count):
                mine = ids[part_of == p]
                self.sets[(p, mode)] = (np.ascontiguousarray(mine), np.ascontiguousarray(self.labels_all[mine]))


def compare_batches(got, want, ctx=""):
    for k in KEYS_EXACT:
        g, w = got[k], want[k]
        assert g.shape == w.shape, f"{ctx}{k}: shape {g.shape} != {w.shape}"
        if not np.array_equal(g, w):
            bad = np.nonzero(g != w)[0]
            raise AssertionError(f"{ctx}{k}: {bad.size} mismatches, first at {bad[0]}: got {g[bad[0]]} want {w[bad[0]]}")
    if "float_features" in want and "float_features" in got:
        g, w = got["float_features"], want["float_features"]
        assert g.shape == w.shape, f"{ctx}features shape {g.shape} != {w.shape}"
        assert np.array_equal(g.view(np.uint32), w.view(np.uint32)), f"{ctx}gathered rows are not byte-identical"


def check_invariants(wl, batch, fanout):
    """Sampler invariants that hold at any size."""
    nc, ec = batch["node_counter"], batch["edge_counter"]
    ids = batch["sampled_ids"]
    assert np.unique(ids).size == ids.size, "sampled_ids has duplicates"
    src, dst = batch["agg_src_ids"], batch["agg_dst_ids"]
    # every emitted edge (dst <- src in the reference's naming: agg_src = neighbour) is a real edge
    deg = wl.indptr[dst.astype(np.int64) + 1] - wl.indptr[dst.astype(np.int64)]
    assert np.all(deg > 0)
    for e in np.random.RandomState(0).choice(src.size, size=min(src.size, 2000), replace=False) if src.size else []:
        row = wl.col[wl.indptr[dst[e]]:wl.indptr[dst[e] + 1]]
        assert src[e] in row
    # localisation
    assert np.array_equal(ids[batch["agg_src_off"]], src)
    assert np.array_equal(ids[batch["agg_dst_off"]], dst)
    # every node is a seed or an endpoint
    b = int(nc[9])
    assert set(ids[b:].tolist()) <= set(src.tolist())
    # per hop: emitted count for a frontier entry = min(fanout, deg)
    hop_num = int(nc[8])
    lo = 0
    frontier = ids[:b]
    for h in range(hop_num):
        hi = int(ec[9 + h + 1])
        fdeg = wl.indptr[frontier.astype(np.int64) + 1] - wl.indptr[frontier.astype(np.int64)]
        assert hi - lo == int(np.minimum(fdeg, fanout[h]).sum())
        frontier = src[lo:hi]
        lo = hi
