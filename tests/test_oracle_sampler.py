"""The C oracle against (i) an independent pure-Python restatement of the reference's sampler on a
hand-made graph, (ii) the committed golden batch, (iii) structural invariants on RMAT graphs."""
import json
import os

import numpy as np
import pytest

from oracle import ffi
from tests.helpers import Workload, check_invariants

GOLD_PATH = os.path.join(os.path.dirname(__file__), "golden", "sampler_tiny.json")


def tiny_graph():
    # 8 vertices: degree 0 (v7), degree 1 (v6), degree < fan-out, degree > fan-out (v0: 9)
    adj = {0: [1, 2, 3, 4, 5, 6, 1, 2, 3], 1: [0, 2], 2: [0, 1, 3], 3: [0, 2, 4, 5], 4: [0, 3],
           5: [0, 3, 6], 6: [5], 7: []}
    indptr = np.zeros(9, dtype=np.int64)
    col = []
    for v in range(8):
        col += adj[v]
        indptr[v + 1] = len(col)
    return indptr, np.array(col, dtype=np.int32)


def py_draw(idx, deg):
    x = pow(48271, idx + 1, 2147483647)
    return int((float(x - 1) / 2147483646.0) * float(deg))


def py_batch(indptr, col, seeds, fanout):
    """SS/engine/operator_impl.cu:27-55,175-296 in plain Python, slot order."""
    ids = list(seeds)
    pos = {v: i for i, v in enumerate(seeds)}
    edges_src, edges_dst = [], []
    frontier = list(seeds)
    ncum, ecum = [len(ids)], [0]
    for f in fanout:
        new_edges = []
        for idx in range(len(frontier) * f):
            s = frontier[idx // f]
            k = idx % f
            deg = int(indptr[s + 1] - indptr[s])
            if k >= deg:
                continue
            d = int(col[indptr[s] + py_draw(idx, deg)])
            if d not in pos:
                pos[d] = len(ids)
                ids.append(d)
            new_edges.append((d, s))
        edges_src += [e[0] for e in new_edges]
        edges_dst += [e[1] for e in new_edges]
        frontier = [e[0] for e in new_edges]
        ncum.append(len(ids))
        ecum.append(len(edges_src))
    return ids, edges_src, edges_dst, [pos[v] for v in edges_src], [pos[v] for v in edges_dst], ncum, ecum


def oracle_batch(indptr, col, seeds, fanout, batch_size, counter=0, mode=1):
    g = ffi.OracleGraph(1, indptr, col)
    p = ffi.OraclePool(indptr.size - 1, batch_size, fanout)
    p.run_batch(g, None, None, seeds, None, batch_size, counter, mode, False)
    out = p.read_batch()
    p.close()
    return out


@pytest.mark.parametrize("fanout", [[3, 2], [25, 10], [2, 2, 2], [1], [4, 1, 3]])
def test_against_python_restatement(fanout):
    indptr, col = tiny_graph()
    seeds = np.array([0, 6, 7, 3], dtype=np.int32)
    got = oracle_batch(indptr, col, seeds, fanout, 4)
    ids, es, ed, so, do, ncum, ecum = py_batch(indptr, col, seeds.tolist(), fanout)
    assert got["sampled_ids"].tolist() == ids
    assert got["agg_src_ids"].tolist() == es and got["agg_dst_ids"].tolist() == ed
    assert got["agg_src_off"].tolist() == so and got["agg_dst_off"].tolist() == do
    h = len(fanout)
    assert got["node_counter"][9:9 + h + 1].tolist() == ncum
    assert got["edge_counter"][9:9 + h + 1].tolist() == ecum
    assert got["node_counter"][8] == h


def test_counter_trace_two_hops():
    # SURVEY.md A.2 worked trace
    indptr, col = tiny_graph()
    seeds = np.array([0, 1, 2, 3], dtype=np.int32)
    got = oracle_batch(indptr, col, seeds, [3, 2], 4)
    nc, ec = got["node_counter"], got["edge_counter"]
    B, U1, U2 = nc[9], nc[10] - nc[9], nc[11] - nc[10]
    E1, E2 = ec[10], ec[11] - ec[10]
    assert B == 4 and nc[0] == B + U1 and nc[1] == U2 and nc[7] == B + U1 + U2 and nc[6] == 0
    assert ec[0] == E1 and ec[1] == E2 and ec[2] == 0 and ec[9] == 0


def test_golden_batch():
    gold = json.load(open(GOLD_PATH))
    indptr, col = tiny_graph()
    for case in gold["cases"]:
        got = oracle_batch(indptr, col, np.array(case["seeds"], dtype=np.int32), case["fanout"],
                           case["batch_size"], case["counter"])
        for k in ("sampled_ids", "agg_src_off", "agg_dst_off", "node_counter", "edge_counter"):
            assert got[k].tolist() == case[k], (case["fanout"], k)


def test_partial_and_empty_batches():
    indptr, col = tiny_graph()
    seeds = np.array([0, 1, 2, 3, 4], dtype=np.int32)
    # counter 1 with batch 4: size = 5 - 4 = 1; the kernel indexes all_ids at size*counter + idx (reference quirk)
    got = oracle_batch(indptr, col, seeds, [2], 4, counter=1)
    assert got["node_counter"][9] == 1 and got["sampled_ids"][0] == seeds[1]
    # counter 2: size = 5 - 8 < 0 -> nothing is sampled
    got = oracle_batch(indptr, col, seeds, [2], 4, counter=2)
    assert got["edge_counter"][10] == 0 and got["sampled_ids"].size == 0


@pytest.mark.parametrize("scale,fanout,batch", [(10, [25, 10], 64), (12, [15, 10, 5], 128), (11, [5], 1000)])
def test_invariants_rmat(scale, fanout, batch):
    wl = Workload(scale=scale, edge_factor=8, dim=0)
    ids, _ = wl.sets[(0, 0)]
    g = ffi.OracleGraph(1, wl.indptr, wl.col)
    p = ffi.OraclePool(wl.N, batch, fanout)
    for counter in range(2):
        p.run_batch(g, None, None, ids, None, batch, counter, 0, False)
        check_invariants(wl, p.read_batch(), fanout)
    p.close()


def test_position_map_cleared_in_train_mode_only():
    indptr, col = tiny_graph()
    g = ffi.OracleGraph(1, indptr, col)
    p = ffi.OraclePool(8, 4, [3])
    seeds = np.array([0, 1, 2, 3], dtype=np.int32)
    p.run_batch(g, None, None, seeds, None, 4, 0, 0, False)
    pm = np.ctypeslib.as_array(p.p.contents.position_map, shape=(8,))
    assert not pm.any()                       # ClearPosMap, operator_impl.cu:542-548
    p.run_batch(g, None, None, seeds, None, 4, 0, 1, False)
    assert pm.any()                           # valid mode leaves it (reference behaviour)
    p.close()
