"""Everything about the first real multi-GPU run that a 1-GPU box CAN find out:
  * the driver's own SCALE command line (`python -m torch.distributed.run ... bench.py --gpus 2 ...`) with both
    ranks on the one GPU and gloo for the collectives -- default layout and --stripe (one clique over the ranks);
  * the RCCL calls themselves, executed once at N = 1 (`--force-dist`, backend nccl);
  * the reference's deployment -- ONE server process, a host thread per GPU (SS/engine/server.cu:95-103,122-130),
    caches striped over a clique of two (cache_agg_mode 1) -- on two LOGICAL GPUs, serving two trainer processes,
    every batch compared with the oracle."""
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import pytest

from tests.gpu_harness import CpuSide
from tests.helpers import Workload
from tests.server_proc import start_server

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--scale", "18", "--group", "8", "--presc-steps", "8", "--cpu-seconds", "0", "--no-boundary", "--min-seconds", "0.1",
         "--steps", "2", "--warmup", "1", "--cache-memory", str(64 << 20)]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _last_json(text):
    lines = [ln for ln in text.splitlines() if ln.startswith("{")]
    assert lines, text[-3000:]
    return json.loads(lines[-1])


@pytest.mark.parametrize("stripe", [False, True], ids=["replicated-caches", "striped-clique"])
def test_scale_command_line_two_ranks_on_one_gpu(hip, stripe):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo",
           "--force-device", "0"] + SMALL + (["--stripe"] if stripe else [])
    res = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-4000:]
    d = _last_json(res.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0 and d["steps"] == 2
    assert d["roofline"]["frac"] > 0 and d["roofline"]["launches"] == 2
    assert ("striped" in d["config"]["parallelism"]) == stripe
    assert "cpu_baseline" not in d                      # N = 1 only
    # the line is its own evidence of what ran on how many ranks: the world size as an all-reduce of
    # ones saw it, the one collective of the path timed, and what every rank measured for itself
    col = d["collective"]
    assert col["backend"] == "gloo" and col["world_size_seen_by_all_reduce"] == 2
    assert col["hotness_all_reduce_ms"] > 0 and col["hotness_all_reduce_bytes"] == 2 * (1 << 18) * 8
    assert [r["rank"] for r in d["per_rank"]] == [0, 1]
    for r in d["per_rank"]:
        assert r["pci_bus_id"] and r["edges_per_sec"] > 0 and 0 < r["gather_roofline_frac"] < 1
    assert abs(sum(r["edges_per_sec"] for r in d["per_rank"]) / d["value"] - 1) < 0.5      # (own clocks vs the slowest rank's)
    legs = [d] if stripe else [d["striped"], d["striped_replica"]]
    for leg in legs:                                     # the striped clique: rows by where the gather read them
        assert "striped" in (leg["config"]["parallelism"] if stripe else leg["parallelism"]) and leg["value"] > 0
        assert leg["roofline"]["frac"] > 0
        for r in leg["per_rank"]:
            assert r["rows_from_peer_stripes"] + r["rows_from_own_stripe"] + r["rows_from_local_replica"] <= r["rows_gathered"]
            assert r["peer_bytes_per_region_computed"] == r["rows_from_peer_stripes"] * 128 * 4
    if not stripe:
        # both ranks sit on ONE GPU here, so a "peer" stripe is local HBM: the striped legs must run at about the headline's
        # rate.  (They once ran 50 x slower: the row-source counting was switched off on the host, but the hipGraphs captured
        # during the counting pass kept the counters' address and went on counting -- one contended atomic per row.)
        assert d["striped"]["value"] > 0.25 * d["value"] and d["striped_replica"]["value"] > 0.25 * d["value"]
        plain, repl = d["striped"]["per_rank"], d["striped_replica"]["per_rank"]
        assert d["striped_replica"]["hot_row_replica_rows"] > 0 and d["striped"]["hot_row_replica_rows"] == 0
        for a, b in zip(plain, repl):                    # the replica takes hit rows away from the stripes, peers' included
            assert a["rows_from_peer_stripes"] > 0 and a["rows_from_local_replica"] == 0 and b["rows_from_local_replica"] > 0
            assert b["rows_from_peer_stripes"] < a["rows_from_peer_stripes"]
        # the owner-bucketed bulk transfer (peer_gather = bulk): the same clique, the peers' rows pushed by their owners -- across
        # two PROCESSES here (IPC handles of lane arenas and request lists), the same edges, exactly the rows `striped` read from peers
        bulk = d["striped_bulk"]
        assert bulk["peer_gather"] == "bulk" and bulk["value"] > 0 and abs(bulk["value"] * bulk["ms_per_step"] / (d["striped"]["value"] * d["striped"]["ms_per_step"]) - 1) < 1e-6
        for a, b in zip(plain, bulk["per_rank"]):
            assert b["bulk"]["rows_pushed_into_me_per_region"] == a["rows_from_peer_stripes"] and b["bulk"]["phase_b_s_per_group"] > 0
    if not stripe:
        # tools/scale_check.py (what will judge the first real SCALE record) reads this line: on two ranks that share ONE GPU over gloo
        # it must flag exactly the three things that are not as on an 8-GPU node -- the PCI bus ids are not distinct, the all-reduce was
        # not the library's RCCL call, the rows "read from peers" crossed no xGMI link -- and pass everything else (world size, per-rank
        # entries, no peer traffic on the headline leg)
        chk = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "scale_check.py"), "-"], input=json.dumps(d), cwd=ROOT,
                             stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=60)
        fails = [ln for ln in chk.stdout.splitlines() if ln.startswith("FAIL")]
        assert chk.returncode == 1 and fails and all(("distinct PCI bus ids" in ln or "issued by" in ln or "xGMI read" in ln) for ln in fails), chk.stdout
        assert sum(ln.startswith("PASS") for ln in chk.stdout.splitlines()) >= 4, chk.stdout
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + SMALL, cwd=ROOT,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert one.returncode == 0, one.stderr[-4000:]
    d1 = _last_json(one.stdout)
    assert d["batches_per_step"] == d1["batches_per_step"] == 8
    # edges_per_step is rank 0's count: same batch shape as the 1-GPU run (different seeds, so only roughly equal)
    assert 0.7 < d["edges_per_step"] / d1["edges_per_step"] < 1.4


def test_scale_command_line_four_ranks_on_one_gpu(hip):
    """The same command line at N = 4 (a GPU box admits six processes on its card): the caches striped over a clique of FOUR ranks
    (cache_agg_mode 2), stripes exchanged as IPC handles between four processes, the bulk leg's arenas mapped from file descriptors
    across four processes -- every leg completes and agrees with the others on what was gathered."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "4", "--backend", "gloo", "--force-device", "0"] + SMALL
    res = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-4000:]
    d = _last_json(res.stdout)
    assert d["n_gpus"] == 4 and d["value"] > 0 and "extra_legs_error" not in d
    assert d["collective"]["world_size_seen_by_all_reduce"] == 4 and [r["rank"] for r in d["per_rank"]] == [0, 1, 2, 3]
    for k in ("striped", "striped_replica", "striped_bulk"):
        assert d[k]["value"] > 0 and "cache_agg_mode 2" in d[k]["parallelism"] and len(d[k]["per_rank"]) == 4, k
    for a, b, r in zip(d["striped"]["per_rank"], d["striped_bulk"]["per_rank"], d["striped_replica"]["per_rank"]):
        assert a["rows_from_peer_stripes"] > 0 and b["bulk"]["rows_pushed_into_me_per_region"] == a["rows_from_peer_stripes"]
        assert r["rows_from_peer_stripes"] < a["rows_from_peer_stripes"]
    # three quarters of the hit rows of a 4-way striped clique live on the other members
    for a in d["striped"]["per_rank"]:
        assert a["rows_from_peer_stripes"] > 2 * a["rows_from_own_stripe"] * 0.8


@pytest.mark.parametrize("how", ["raise", "exit", "sigterm", "hang"])
def test_headline_line_survives_a_failing_extra_leg(hip, how):
    """N > 1: the striped legs run after the headline leg; if one of them ends rank 0 -- a Python error, a native exit() of the
    library, SIGTERM from the launcher, a hang -- the headline leg's ONE line must still be printed, with a note (bench.py OneLine)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo",
           "--force-device", "0", "--fail-extra-leg", how, "--extra-legs-deadline", "8"] + SMALL
    res = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, (res.stdout[-2000:], res.stderr[-2000:])
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["roofline"]["frac"] > 0
    assert "extra_legs_error" in d and "striped" not in d


@pytest.mark.parametrize("how", ["exit", "sigterm", "hang", "raise"])
def test_headline_line_survives_a_failing_leg_at_n1(hip, how):
    """N = 1 (round 6): the legs behind the headline leg -- cold regather, other shapes, boundary, CPU baseline, traffic children -- run
    with the guard armed: a native exit(), SIGTERM, a hang still print the headline's line (with a note); a Python error inside the
    cold leg is caught where it happens and costs only that leg's fields."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--cold-leg", "--fail-extra-leg", how, "--post-legs-deadline", "8"] + SMALL
    res = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, (res.stdout[-2000:], res.stderr[-2000:])
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["roofline"]["frac"] > 0 and d["roofline"]["unique_row_frac"] > 0
    if how == "raise":
        assert d["roofline"]["cold"] is None and "requested by --fail-extra-leg" in d["roofline"]["cold_note"] and "extra_legs_error" not in d
    else:
        assert "extra_legs_error" in d and "cold" not in d["roofline"]


def test_rccl_calls_execute_at_n1(hip):
    """bench.py --force-dist: torch.distributed over the nccl backend (= RCCL) with world size 1; the hotness all-reduce,
    the MIN/MAX reductions and the barriers all go through RCCL once."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--backend", "nccl"] + SMALL,
                         cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-4000:]
    d = _last_json(res.stdout)
    assert d["n_gpus"] == 1 and d["value"] > 0


@pytest.mark.parametrize("partition_file,peer_gather,disk", [(False, "direct", False), (True, "direct", False), (False, "bulk", False),
                                                             (False, "direct", True)],
                         ids=["modulo", "partition-file", "modulo-bulk-peer-gather", "disk-mode-hybrid-tier"])
def test_one_server_process_two_gpus_thread_per_gpu(hip, tmp_path, partition_file, peer_gather, disk):
    """`sampling_server 2 1 5 3`: GPUServer with two runners (a host thread each), PreSC on both, hotness summed over the
    clique, caches striped Kg = 2, two pipe-slot sets, two trainer processes.  Logical GPU 1 shares the box's one GPU.
    With a `partition` file in the dataset directory (the reference's xtrapulp output, storage_management.cu:165-183)
    training seeds go to the GPU the file names -- an uneven split, entries >= 2 dropped -- while validation and testing
    seeds stay on id % 2.  peer_gather = bulk: the rows a runner needs from the OTHER member's stripe are listed per owner and pushed
    by kernels on the owner's device (LegionTuning.peer_gather, pipeline.hip) instead of being loaded through peer pointers.
    disk: `sampling_server 2 1 5 3 --disk` with the fifteen-field meta_config: no clique at all -- every GPU builds the hybrid
    CPU-cache / GPU-cache tier from its OWN counters (UnifiedCache::HybridInit, SS/cache/cache.cu:626-643), one CPU cache per GPU."""
    scale, D, B, fanout, epoch, cache_memory = 11, 24, 40, [5, 3], 2, 40_000
    part = None
    if partition_file:
        part = np.random.RandomState(5).choice(3, size=1 << scale, p=[0.55, 0.35, 0.10]).astype(np.int32)
    wl = Workload(scale=scale, edge_factor=8, dim=D, n_seeds=700, n_valid=130, n_test=70, partition_count=2, partition=part)
    N = wl.N
    train, valid, test = wl.train, wl.valid, wl.test                           # the split before it is partitioned
    ds = str(tmp_path / "ds") + "/"
    os.makedirs(ds)
    if partition_file:
        part.tofile(ds + "partition")
        assert wl.sets[(0, 0)][0].size != wl.sets[(1, 0)][0].size and wl.sets[(0, 0)][0].size + wl.sets[(1, 0)][0].size < 700
    wl.indptr.astype(np.int64).tofile(ds + "edge_src"); wl.col.astype(np.int32).tofile(ds + "edge_dst")
    wl.features.astype(np.float32).tofile(ds + "features"); wl.labels_all.astype(np.int32).tofile(ds + "labels")
    train.tofile(ds + "trainingset"); valid.tofile(ds + "validationset"); test.tofile(ds + "testingset")
    work = tmp_path / "run"
    work.mkdir()
    cpu_cap, gpu_cap = 140, 110
    (work / "meta_config").write_text("{} {} {} {} {} {} {} {} {} {}".format(
        ds, B, N, wl.col.size, D, train.size, valid.size, test.size, cache_memory, epoch) +
        (" 0 0 0 {} {}".format(cpu_cap, gpu_cap) if disk else ""))
    ns = f"_m{os.getpid()}"
    env = dict(os.environ, LEGION_IPC_NAMESPACE=ns, LEGION_PEER_GATHER=peer_gather)
    server, log = start_server([os.path.join(ROOT, "legion_amd", "bin", "sampling_server"), "2", "1"] + [str(f) for f in fanout] +
                               (["--disk"] if disk else []), work, env, work / "server.log")
    trainers = []
    try:
        for dev in range(2):
            trainers.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "fake_trainer.py"), str(dev), str(D),
                                              str(epoch), str(tmp_path / f"t{dev}.npz")], env=env, cwd=ROOT,
                                             stdout=open(tmp_path / f"t{dev}.log", "w"), stderr=subprocess.STDOUT,
                                             stdin=subprocess.DEVNULL))
        for dev, t in enumerate(trainers):
            t.wait(timeout=240)
            assert t.returncode == 0, open(tmp_path / f"t{dev}.log").read()[-3000:]
        server.wait(timeout=120)
        assert server.returncode == 0, open(work / "server.log").read()[-3000:]
        text = open(work / "server.log").read()
        assert ("pushed by their owners (peer_gather = bulk)" in text) == (peer_gather == "bulk"), text[-2000:]
        if disk:
            assert "Finish initializing cache" in text and "Alpha:" not in text and "GPU Cache Capacity: %d" % gpu_cap in text
        else:
            assert "leader loop over peer pointers (its logical GPUs share physical devices)" in text      # one GPU in the box: no RCCL communicator

        # ---- the oracle: the same two-GPU server in one address space --------------------------------
        from oracle import ffi
        import ctypes
        L = ffi.load()
        st = ffi.Steps()
        arr = lambda v: (ctypes.c_int32 * 2)(*v)
        n_of = lambda mode: [wl.sets[(p, mode)][0].size for p in range(2)]
        L.lgo_coordinate(ctypes.byref(st), 2, arr(n_of(0)), arr(n_of(1)), arr(n_of(2)), B, epoch)
        max_bs = max([B] + [st.valid_bs[p] for p in range(2)] + [st.test_bs[p] for p in range(2)])
        cpu = CpuSide(wl, max_bs, fanout)
        for p in range(2):
            for it in range(st.train_step):
                cpu.run(p, it, 0, is_presc=True, batch_size=B)
        if disk:
            cpu.Kg, cpu.caches = 1, []
            for p in range(2):
                c = ffi.OracleCache(wl.N, wl.D, 1, p)
                c.hybrid_init(cpu.node_access[p], wl.features, cpu_cap, gpu_cap)
                cpu.caches.append(c)
            assert not np.array_equal(cpu.caches[0].arr("QF", np.int32)[:50], cpu.caches[1].arr("QF", np.int32)[:50])    # two orders
        else:
            cpu.build_cache(1, cache_memory=cache_memory, train_step=st.train_step)
        total = L.lgo_max_step(ctypes.byref(st))
        H = len(fanout)
        hits = 0
        for p in range(2):
            got = np.load(tmp_path / f"t{p}.npz")
            assert got["steps"].tolist() == [st.train_step, st.valid_step, st.test_step]
            for gb in range(total):
                mode = L.lgo_current_mode(ctypes.byref(st), gb)
                it = L.lgo_local_batch_id(ctypes.byref(st), gb)
                bs = L.lgo_current_batchsize(ctypes.byref(st), p, mode)
                want = cpu.run(p, it, mode, batch_size=bs)
                nc, ec = want["node_counter"], want["edge_counter"]
                assert np.array_equal(got[f"b{gb}_ids"], want["sampled_ids"]), f"gpu {p} batch {gb}"
                assert np.array_equal(got[f"b{gb}_labels"], want["labels"])
                assert np.array_equal(got[f"b{gb}_feats"], want["float_features"].view(np.uint32)), f"gpu {p} batch {gb} rows"
                for k, h in enumerate(range(H, 0, -1)):
                    n_e = int(ec[9 + h])
                    assert np.array_equal(got[f"b{gb}_src{k}"], want["agg_src_off"][:n_e])
                    assert np.array_equal(got[f"b{gb}_dst{k}"], want["agg_dst_off"][:n_e])
                exp = []
                for h in range(H, 0, -1):
                    exp += [int(nc[9 + h]), int(nc[9 + h - 1])]
                assert got[f"b{gb}_sizes"].tolist() == exp
                hits += int((want["cache_search_buffer"] >= 0).sum()) if "cache_search_buffer" in want else 0
        cpu.close()
    finally:
        for t in trainers:
            if t.poll() is None:
                t.kill()
        if server.poll() is None:
            server.kill()
        log.close()
        for name in os.listdir("/dev/shm"):
            if name.endswith(ns):
                os.unlink(os.path.join("/dev/shm", name))
