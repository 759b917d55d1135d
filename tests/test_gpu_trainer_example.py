"""A real consumer on the other side of the boundary: `examples/graphsage_torch.py` (the call sequence of
training_backend/legion_graphsage.py:72-128,148-178 with DGLBlock / SAGEConv('mean') written in plain PyTorch) trains
against the `sampling_server` binary on a task that can ONLY be learnt through the blocks: a seed's label is the group
all of its neighbours belong to, its own features say nothing about it.  If ids, feature rows, labels, the COO blocks or
the block sizes were misaligned in any way the accuracy would stay at chance (1/4); it reaches ~1."""
import json
import os
import subprocess
import sys
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_graphsage_learns_a_neighbour_only_task_through_the_boundary(hip, tmp_path):
    rng = np.random.RandomState(4)
    N, C, D, B, fanout, epoch = 4096, 4, 16, 64, [6, 4], 5
    group = rng.randint(0, C, N)                              # what a vertex's OWN features say
    target = rng.randint(0, C, N)                             # the group all of its neighbours are drawn from = its label
    members = [np.nonzero(group == c)[0] for c in range(C)]
    deg = rng.randint(1, 13, N).astype(np.int64)
    indptr = np.zeros(N + 1, dtype=np.int64)
    np.cumsum(deg, out=indptr[1:])
    col = np.empty(indptr[-1], dtype=np.int32)
    for v in range(N):
        col[indptr[v]:indptr[v + 1]] = rng.choice(members[target[v]], size=deg[v])
    feats = rng.randn(N, D).astype(np.float32)
    feats[:, :C] = 0
    feats[np.arange(N), group] = 1.0
    perm = rng.permutation(N).astype(np.int32)
    train, valid, test = perm[:3000], perm[3000:3500], perm[3500:4000]

    ds = str(tmp_path / "ds") + "/"
    os.makedirs(ds)
    indptr.tofile(ds + "edge_src"); col.tofile(ds + "edge_dst"); feats.tofile(ds + "features")
    target.astype(np.int32).tofile(ds + "labels")
    train.tofile(ds + "trainingset"); valid.tofile(ds + "validationset"); test.tofile(ds + "testingset")
    work = tmp_path / "run"
    work.mkdir()
    (work / "meta_config").write_text("{} {} {} {} {} {} {} {} {} {}".format(
        ds, B, N, col.size, D, train.size, valid.size, test.size, 100_000, epoch))
    ns = f"_g{os.getpid()}"
    env = dict(os.environ, LEGION_IPC_NAMESPACE=ns)
    from tests.server_proc import start_server
    server, log = start_server([os.path.join(ROOT, "legion_amd", "bin", "sampling_server"), "1", "0"] + [str(f) for f in fanout],
                               work, env, work / "server.log")
    trainer = None
    try:
        report = tmp_path / "report.json"
        trainer = subprocess.Popen([sys.executable, os.path.join(ROOT, "examples", "graphsage_torch.py"), "--device", "0",
                                    "--features_num", str(D), "--hidden_dim", "32", "--class_num", str(C), "--hops_num", "2",
                                    "--drop_rate", "0", "--lr", "0.01", "--epoch", str(epoch), "--report", str(report)],
                                   env=env, cwd=ROOT, stdout=open(tmp_path / "trainer.log", "w"), stderr=subprocess.STDOUT,
                                   stdin=subprocess.DEVNULL)
        trainer.wait(timeout=500)
        assert trainer.returncode == 0, open(tmp_path / "trainer.log").read()[-3000:]
        server.wait(timeout=120)
        assert server.returncode == 0, open(work / "server.log").read()[-3000:]
        rep = json.load(open(report))
        hist = rep["history"]
        assert len(hist) == epoch
        assert hist[-1]["train_loss"] < 0.5 * hist[0]["train_loss"], hist
        assert hist[-1]["valid_acc"] > 0.95 and rep["test_acc"] > 0.95, rep      # chance = 0.25
    finally:
        if trainer is not None and trainer.poll() is None:
            trainer.kill()
        if server.poll() is None:
            server.kill()
        log.close()
        for name in os.listdir("/dev/shm"):
            if name.endswith(ns):
                os.unlink(os.path.join("/dev/shm", name))
