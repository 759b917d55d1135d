"""The N > 1 path on CPU with gloo, world_size 2 and 8: seeds are sharded id % P (storage_management.cu:178),
each rank runs its own PreSC epoch, hotness is all-reduced (the path's only collective; RCCL on the
GPUs), and every rank derives the same cache order / capacities while sampling only its own seeds."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.helpers import Workload


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import ffi
    wl = Workload(scale=10, edge_factor=8, dim=8, n_seeds=800, partition_count=world)
    fanout, batch = [5, 3], 32
    ids, labels = wl.sets[(rank, 0)]
    assert np.all(ids % world == rank)
    g = ffi.OracleGraph(1, wl.indptr, wl.col)
    pool = ffi.OraclePool(wl.N, batch, fanout, ffi.num_ids_for(batch, fanout), wl.D)
    node_acc = np.zeros(wl.N, dtype=np.uint64)
    edge_acc = np.zeros(wl.N, dtype=np.uint64)
    steps = (ids.size - 1) // batch
    t_steps = torch.tensor([steps])
    dist.all_reduce(t_steps, op=dist.ReduceOp.MIN)            # train_step = min over partitions (ipc_service.cu:73-82)
    steps = int(t_steps.item())
    max_ids = 0
    edges = 0
    for it in range(steps):
        edges += pool.run_batch(g, None, None, ids, labels, batch, it, 0, True, node_acc, edge_acc)
        max_ids = max(max_ids, int(pool.read_batch()["node_counter"][7]))
    # the hotness all-reduce (uint64 counters viewed as int64, exactly what bench.py does with RCCL)
    tn = torch.from_numpy(node_acc.view(np.int64))
    te = torch.from_numpy(edge_acc.view(np.int64))
    local_sum = int(node_acc.sum())
    dist.all_reduce(tn)
    dist.all_reduce(te)
    cache = ffi.OracleCache(wl.N, wl.D, 1, 0)
    cache.candidate_selection([node_acc], [edge_acc])
    cache.cost_model(200_000, wl.indptr, (0, 0), [max_ids], steps)
    cache.fill_up(wl.features, wl.indptr, wl.col)
    g.attach_cache(cache)
    qf = cache.arr("QF", np.int32)
    pool.run_batch(g, cache, wl.features, ids, labels, batch, 0, 0, False)
    b = pool.read_batch()
    q.put((rank, steps, local_sum, int(node_acc.sum()), qf[:50].tolist(), cache.node_capacity, cache.edge_capacity,
           edges, b["sampled_ids"][:batch].tolist(), int((b["cache_search_buffer"] >= 0).sum())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_ranks_gloo(world):
    """world 2, and 8 -- the target machine's width (a GPU box admits at most 6 processes on its one card, so the
    width-8 rehearsal of the sharding and of the hotness reduce lives here, on the oracle)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    r0 = res[0]
    assert r0[1] >= 1
    assert r0[3] == sum(r[2] for r in res)                      # all-reduced hotness = sum of the ranks'
    seen = set()
    for rank, r in enumerate(res):
        assert r[0] == rank and r[1] == r0[1]                   # same train_step everywhere
        assert r[3] == r0[3] and r[4] == r0[4]                  # the same sums, hence the identical cache order on every rank
        assert (r[5], r[6]) == (r0[5], r0[6])                   # identical capacities
        assert all(v % world == rank for v in r[8])             # seed shards by id % P ...
        assert seen.isdisjoint(r[8])                            # ... and disjoint
        seen.update(r[8])
        assert r[7] > 0
    assert r0[9] > 0
