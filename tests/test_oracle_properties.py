"""Property tests (hypothesis) of the oracle on random small graphs: it equals the pure-Python
restatement and satisfies the sampler invariants for arbitrary graphs, seeds and fan-outs."""
import numpy as np
from hypothesis import given, settings, strategies as st

from tests.test_oracle_sampler import oracle_batch, py_batch


@st.composite
def graphs(draw):
    n = draw(st.integers(2, 40))
    degs = draw(st.lists(st.integers(0, 12), min_size=n, max_size=n))
    col = []
    indptr = [0]
    for v in range(n):
        for _ in range(degs[v]):
            col.append(draw(st.integers(0, n - 1)))
        indptr.append(len(col))
    seeds = draw(st.lists(st.integers(0, n - 1), min_size=1, max_size=min(n, 10), unique=True))
    # one to SIX hops (the counter block holds no more, operator_impl.cu:67,81-82); deep lists get small fan-outs to bound the slots
    hops = draw(st.integers(1, 6))
    fanout = draw(st.lists(st.integers(1, 6 if hops <= 3 else 3), min_size=hops, max_size=hops))
    return np.array(indptr, dtype=np.int64), np.array(col, dtype=np.int32), seeds, fanout


@settings(max_examples=120, deadline=None)
@given(graphs())
def test_oracle_equals_python_restatement(case):
    indptr, col, seeds, fanout = case
    got = oracle_batch(indptr, col, np.array(seeds, dtype=np.int32), fanout, len(seeds))
    ids, es, ed, so, do, ncum, ecum = py_batch(indptr, col, seeds, fanout)
    assert got["sampled_ids"].tolist() == ids
    assert got["agg_src_ids"].tolist() == es and got["agg_dst_ids"].tolist() == ed
    assert got["agg_src_off"].tolist() == so and got["agg_dst_off"].tolist() == do
    h = len(fanout)
    assert got["node_counter"][9:9 + h + 1].tolist() == ncum
    assert got["edge_counter"][9:9 + h + 1].tolist() == ecum


@settings(max_examples=120, deadline=None)
@given(graphs())
def test_sampler_invariants(case):
    indptr, col, seeds, fanout = case
    got = oracle_batch(indptr, col, np.array(seeds, dtype=np.int32), fanout, len(seeds))
    ids, src, dst = got["sampled_ids"], got["agg_src_ids"], got["agg_dst_ids"]
    assert len(set(ids.tolist())) == ids.size                              # no duplicates
    assert ids[:len(seeds)].tolist() == seeds                              # seeds first, in order
    for s, d in zip(src.tolist(), dst.tolist()):                           # every edge is a real edge
        assert s in col[indptr[d]:indptr[d + 1]].tolist()
    assert np.array_equal(ids[got["agg_src_off"]], src) and np.array_equal(ids[got["agg_dst_off"]], dst)
    assert set(ids[len(seeds):].tolist()) <= set(src.tolist())
    ec = got["edge_counter"]
    frontier = np.array(seeds)
    lo = 0
    for h, f in enumerate(fanout):                                         # with replacement, capped at degree
        hi = int(ec[9 + h + 1])
        deg = indptr[frontier + 1] - indptr[frontier]
        assert hi - lo == int(np.minimum(deg, f).sum())
        frontier = src[lo:hi]
        lo = hi
