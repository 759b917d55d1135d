"""Synthetic workload generators (numpy restatement of the device generators)."""
import numpy as np

from legion_amd import synth


def test_rmat_csr_is_valid_and_deterministic():
    ip, col = synth.rmat_csr_numpy(10, 8, 20231)
    ip2, col2 = synth.rmat_csr_numpy(10, 8, 20231)
    assert np.array_equal(ip, ip2) and np.array_equal(col, col2)
    assert ip[0] == 0 and ip[-1] == col.size == 1024 * 8 and np.all(np.diff(ip) >= 0)
    assert col.min() >= 0 and col.max() < 1024
    src = np.repeat(np.arange(1024), np.diff(ip))
    assert not np.any(src == col)                       # no self loops
    deg = np.diff(ip)
    assert deg.max() > 20 * deg.mean()                  # skewed, as RMAT should be


def test_features_are_exact_and_bounded():
    f = synth.features_numpy(0, 100, 16, 7)
    assert f.dtype == np.float32 and f.min() >= -1.0 and f.max() < 1.0 and not np.isnan(f).any()
    assert np.array_equal(synth.features_numpy(40, 10, 16, 7), f[40:50])
    assert np.array_equal(synth.feature_rows_numpy([3, 99, 3], 16, 7), f[[3, 99, 3]])
    assert len(np.unique(f.view(np.uint32))) > 1500     # rows are distinguishable


def test_seed_ids_are_a_permutation_prefix():
    s = synth.seed_ids(1 << 12, 1 << 12, 11)
    assert np.array_equal(np.sort(s), np.arange(1 << 12))
    assert np.array_equal(synth.seed_ids(1 << 12, 100, 11), s[:100])
    t = synth.seed_ids(1000, 300, 11)
    assert np.unique(t).size == 300 and t.max() < 1000


def test_label_scrambling_is_a_bijection_that_keeps_the_graph():
    for scale in (1, 5, 10, 13):
        x = np.arange(1 << scale)
        y = synth.scramble_labels_numpy(x, scale, synth.SCRAMBLE_KEY)
        assert np.array_equal(np.sort(y), x)
    ip, col = synth.rmat_csr_numpy(10, 8, 20231)
    ips, cols = synth.rmat_csr_numpy(10, 8, 20231, scramble=True)
    perm = synth.scramble_labels_numpy(np.arange(1024), 10, synth.SCRAMBLE_KEY).astype(np.int64)
    deg, degs = np.diff(ip), np.diff(ips)
    assert np.array_equal(degs[perm], deg)                       # vertex v became perm[v], same degree
    for v in (0, 1, 2, 77, 1023):                                # and the same (relabelled) neighbours
        assert np.array_equal(np.sort(perm[col[ip[v]:ip[v + 1]]]), np.sort(cols[ips[perm[v]]:ips[perm[v] + 1]]))
    hubs = np.argsort(-degs)[:32]
    assert hubs.max() > 512                                      # hubs are no longer the low ids
