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
