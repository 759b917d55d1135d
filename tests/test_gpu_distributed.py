"""A clique spread over PROCESSES: `world` ranks (here all on the one GPU of the box, gloo for the
collectives), each owning member `rank` of a Kg = world clique.  Hotness is all-reduced, every rank
builds its stripe of the feature cache and of the topology cache, the stripes are exchanged as IPC
handles, and every rank's mini-batches -- whose cache hits now mostly read PEER memory -- must equal the
oracle's for a Kg = world clique bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    try:
        sys.path.insert(0, ROOT)
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        import torch
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.cuda.set_device(0)
        from legion_amd import engine
        from tests.gpu_harness import CpuSide
        from tests.helpers import Workload, compare_batches
        engine.set_local_device(rank)          # every rank: logical GPU `rank`, physically the box's one GPU
        wl = Workload(scale=11, edge_factor=8, dim=32, n_seeds=1600, partition_count=world)
        fanout, batch = [5, 4], 64
        dev = torch.device("cuda:0")
        indptr, col = torch.from_numpy(wl.indptr).to(dev), torch.from_numpy(wl.col).to(dev)
        feats = torch.from_numpy(wl.features).to(dev)
        graph = engine.GraphStorage(world, indptr, col)
        feature = engine.FeatureStorage(world, feats, wl.N, wl.D)
        for mode in (0, 1):
            ids, labels = wl.sets[(rank, mode)]
            feature.set_ids(rank, mode, ids, labels)
        steps = min((wl.sets[(p, 0)][0].size - 1) // batch for p in range(world))
        cache = engine.UnifiedCache(0, wl.D, steps, world, wl.N)
        cache.init_controller(rank)
        pool = engine.MemoryPool(rank, wl.N, batch, fanout, wl.D)
        pool.alloc_features(pool.num_ids)
        for it in range(steps):
            engine.enqueue_batch(None, graph, feature, cache, pool, batch, it, rank, 0, True, fanout)
        torch.cuda.synchronize()
        # the path's only collective: hotness all-reduce (RCCL on a real node; gloo here)
        dist.all_reduce(cache.array("node_access_time", rank))
        dist.all_reduce(cache.array("edge_access_time", rank))
        mids = [None] * world
        dist.all_gather_object(mids, cache.max_id_num(rank))
        cache.set_peer_max_ids(mids)
        mode_bits = int(np.log2(world))
        cache.candidate_selection(mode_bits, graph, world_reduced=True)
        cap = (90, 40)
        cache.set_capacity(*cap)

        def all_gather_bytes(b):
            out = [None] * world
            dist.all_gather_object(out, b)
            return out

        cache.fill_up_distributed(feature, graph, rank, world, mids, all_gather_bytes)
        dist.barrier()

        # the oracle: the same clique in one address space
        cpu = CpuSide(wl, batch, fanout)
        for p in range(world):
            for it in range(steps):
                cpu.run(p, it, 0, is_presc=True)
        cpu.build_cache(mode_bits, capacity=cap)
        assert np.array_equal(cache.array("QF", rank).cpu().numpy(), cpu.caches[0].arr("QF", np.int32))
        assert np.array_equal(cache.array("node_map", rank).cpu().numpy(), cpu.caches[0].arr("node_map", np.int32))
        remote_hits = 0
        for mode in (0, 1):
            for it in range(2):
                engine.enqueue_batch(None, graph, feature, cache, pool, batch, it, rank, mode, False, fanout)
                torch.cuda.synchronize()
                g, c = engine.read_batch(pool), cpu.run(rank, it, mode)
                compare_batches(g, c, f"rank {rank} mode {mode} batch {it}: ")
                ci = pool.buffer("cache_search_buffer")[:int(g["node_counter"][1])].cpu().numpy()
                assert np.array_equal(ci, c["cache_search_buffer"])
                remote_hits += int(((ci >= 0) & (ci // cap[0] != rank)).sum())
        # ---- the same batches with the remote rows moved by the OWNERS (peer_gather = bulk): lists and lane arenas exchanged as
        #      IPC handles, two host barriers per launch group ------------------------------------------------------------
        G = 3
        pipe = engine.Pipeline(graph, feature, cache, rank, batch, fanout, G, pool.num_ids, use_graph=False, slots=2, arena="shared")
        pipe.bulk_enable()
        mine = pipe.bulk_export()
        # (LegionBulkHandles: five 64-byte IPC handles, cap, slots, member, then arena_kind: 1 = chunks served as file descriptors)
        kind = int(np.frombuffer(mine[336:340], dtype=np.int32)[0])
        assert kind == (0 if os.environ.get("LEGION_ARENA_SCATTER_MB") == "0" else 1), kind
        for r, h in enumerate(all_gather_bytes(mine)):
            if r != rank:
                pipe.bulk_import(h)
        bulk_rows = 0
        for grp in range(2):
            slot = pipe.bulk_phase_a(grp * G)
            dist.barrier()                  # every member's lists are final
            bulk_rows += pipe.bulk_listed(slot)
            pipe.bulk_phase_b(slot)
            dist.barrier()                  # every member's pushes have landed
            for lane in range(G):
                g, c = engine.read_batch(pipe.pools[slot][lane]), cpu.run(rank, grp * G + lane, 0)
                compare_batches(g, c, f"bulk rank {rank} group {grp} lane {lane}: ")
            dist.barrier()                  # nobody's lanes are overwritten before everybody has looked
        assert bulk_rows > 0, f"rank {rank}: no row travelled by the bulk path"
        dist.barrier()                      # peers keep their stripes alive until everyone is done
        pipe.close()
        if kind == 1:
            # the thread that handed this arena's chunks out went with the arena: nobody answers under its name any more
            import socket as _s
            import time as _t
            pid_, tag_ = (int(v) for v in np.frombuffer(mine[344:352], dtype=np.int32))
            assert pid_ == os.getpid()
            gone = False
            for _ in range(100):
                c = _s.socket(_s.AF_UNIX, _s.SOCK_STREAM)
                try:
                    c.connect(f"\0legion_bulk_{pid_}_{tag_}")
                except ConnectionRefusedError:
                    gone = True
                finally:
                    c.close()
                if gone:
                    break
                _t.sleep(0.02)
            assert gone, "the arena's descriptor socket outlived the arena"
        q.put((rank, "ok", remote_hits))
        dist.destroy_process_group()
    except Exception as e:                  # pragma: no cover
        import traceback
        q.put((rank, "fail: " + repr(e) + "\n" + traceback.format_exc(), 0))


@pytest.mark.parametrize("world,arena_mb", [(2, None), (4, None), (2, "0")], ids=["2-ranks", "4-ranks", "2-ranks-plain-arenas"])
def test_clique_across_processes(hip, world, arena_mb, monkeypatch):
    """(the bulk leg's lane arenas: shuffled chunks that the other ranks map from file descriptors -- round 5 -- or, with
    LEGION_ARENA_SCATTER_MB=0, one plain allocation opened through an IPC handle)"""
    if arena_mb is not None:
        monkeypatch.setenv("LEGION_ARENA_SCATTER_MB", arena_mb)
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    for rank, status, remote in res:
        assert status == "ok", f"rank {rank}: {status}"
        assert remote > 0, f"rank {rank} never read a peer's stripe"
