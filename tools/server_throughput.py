"""Throughput of the drop-in boundary itself: the `sampling_server` binary serving a trainer that only
consumes (get_next -> synchronize), on a synthetic dataset written in the reference's file formats.
Usage (GPU box):  python tools/server_throughput.py [--scale 22] [--batch 8000] [--dim 128]
Prints one JSON line: batches/s and sampled edges/s through the IPC hand-off (train mode, one GPU)."""
import argparse, json, os, subprocess, sys, tempfile, time
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "legion_amd", "trainer"))
from legion_amd import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=int, default=22)
    ap.add_argument("--batch", type=str, default="8000", help="one batch size or a comma list (one server run and one JSON line each, same data set)")
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--fanout", type=str, default="25,10")
    ap.add_argument("--train-batches", type=int, default=60)
    ap.add_argument("--cache-memory", type=int, default=1 << 30)
    ap.add_argument("--epochs", type=int, default=1, help="epochs the server runs (python consumer: the timed window spans them all)")
    ap.add_argument("--verify-every", type=int, default=0,
                    help="python consumer: every k-th batch, check on the device that the rows are the generator's rows of the "
                         "batch's ids, that ids are unique and that every edge endpoint indexes a node of the batch (soak test "
                         "of the hand-over: a wrong slot, a stale or half-copied batch fails here)")
    ap.add_argument("--consumer", type=str, default="python", choices=["python", "native"],
                    help="python: the ipc_service extension as a trainer would use it (get_next -> synchronize); "
                         "native: tools/boundary_consumer.c, the wire protocol with nothing else (what the server can hand over)")
    ap.add_argument("--consume", action="store_true",
                    help="python consumer: READ every batch -- one launch per get_next that loads every float of the rows and every entry of "
                         "the outermost COO pair (legion_consume_batch), completed before the batch is released -- instead of only walking "
                         "the protocol")
    ap.add_argument("--no-features-file", action="store_true",
                    help="do not write the `features` file: the server then serves a zero-filled table of the same shape (v2 of the "
                         "reference reads no features either, storage_management.cu:162); a 34 GB file is too slow to write for a "
                         "throughput run at RMAT-26, the traffic is the same")
    ap.add_argument("--handover", type=str, default="", choices=["", "auto", "gather"],
                    help="LEGION_RUNNER_HANDOVER for the server (default: whatever the environment says, i.e. auto)")
    ap.add_argument("--no-views", action="store_true",
                    help="the consumer does not take batches as views of the server's lane arena (a trainer end that knows only the "
                         "reference's slab): LEGION_NO_DIRECT_VIEWS=1 for the python consumer, no `views` flag for the native one")
    ap.add_argument("--modes", type=str, default="",
                    help="comma list of views|slab|gather: one server run per (mode, batch size) over the same data set files.  views = "
                         "server as it starts by default + a consumer that takes views of the lane arena; slab = the same server + a "
                         "consumer that opens only the reference's slab (the server then gathers each batch into the pipe slot); gather = "
                         "LEGION_RUNNER_HANDOVER=gather (that hand-over for everybody, no arena); overrides --handover / --no-views")
    ap.add_argument("--min-timed-batches", type=int, default=0,
                    help="run as many epochs as it takes for the timed window to hold at least this many batches (views mode hands "
                         "over 100 k+ batches/s: a single epoch of a few thousand batches is a window of milliseconds)")
    ap.add_argument("--profile-server", type=str, default="",
                    help="directory: run the server under `rocprofv3 --kernel-trace --output-format csv -d <dir>/<mode>_b<batch>` (its kernel "
                         "trace: per-batch hand-over launches, gaps, what runs beside them)")
    ap.add_argument("--edge-factor", type=int, default=16)
    ap.add_argument("--nodes", type=int, default=0, help="with --edges: synth.csr_device_large(nodes, edges) instead of RMAT-<scale>")
    ap.add_argument("--edges", type=int, default=0)
    ap.add_argument("--watchdog", type=int, default=0, help="seconds after which the tool dumps its stack and the server log and exits")
    a = ap.parse_args()
    if a.watchdog:
        import faulthandler
        faulthandler.dump_traceback_later(a.watchdog, exit=False)
    fanout = [int(x) for x in a.fanout.split(",")]
    batches = [int(x) for x in a.batch.split(",")]
    dev = torch.device("cuda:0")
    N = a.nodes if a.nodes > 0 else 1 << a.scale
    if a.nodes > 0:
        indptr, col = synth.csr_device_large(N, a.edges, 20231, dev)
    else:
        indptr, col = synth.rmat_csr_device(a.scale, a.edge_factor, 20231, dev)
    tb = {b: (a.train_batches if len(batches) == 1 else max(64, min(a.train_batches, a.train_batches * batches[0] // b))) for b in batches}
    if a.min_timed_batches > 0:           # many short epochs: make each a whole number of launch groups (groups never straddle epochs)
        tb = {b: (v // 64 * 64 if v >= 64 else v) for b, v in tb.items()}
    train = synth.seed_ids(N, max(b * tb[b] for b in batches) + 1, 11).astype(np.int32)
    tmp = tempfile.mkdtemp(prefix="legion_srv_", dir="/tmp")
    ds = os.path.join(tmp, "ds") + "/"
    os.makedirs(ds)
    indptr.cpu().numpy().astype(np.int64, copy=False).tofile(ds + "edge_src")
    col.cpu().numpy().astype(np.int32, copy=False).tofile(ds + "edge_dst")
    if not a.no_features_file:
        feats = synth.features_device(N, a.dim, 7, dev)
        feats.cpu().numpy().tofile(ds + "features")
        del feats
    (np.arange(N) % 47).astype(np.int32).tofile(ds + "labels")
    train.tofile(ds + "trainingset")
    E = int(col.numel())
    del indptr, col
    torch.cuda.empty_cache()
    try:
        for mode in (a.modes.split(",") if a.modes else [""]):
            if mode:
                a.handover = mode if mode == "gather" else "auto"
                a.no_views = mode != "views"
                os.environ.pop("LEGION_NO_DIRECT_VIEWS", None)
            for b in batches:
                n_train = b * tb[b] + 1
                train[:b].tofile(ds + "validationset"); train[:b].tofile(ds + "testingset")
                if a.min_timed_batches > 0:
                    a.epochs = max(1, -(-a.min_timed_batches // tb[b]))
                run_one(a, ds, tmp, b, n_train, train, fanout, N, E, mode)
    finally:
        subprocess.call(["rm", "-rf", tmp])


def consume(spec):
    """The Python consumer of ONE server life, in a process of its own (a trainer attaches to one server; several server lives
    attached to and detached from one long-lived process made a device-to-host copy of a freshly opened IPC buffer abort inside the
    HIP runtime now and then): walks the schedule through the `ipc_service` module, times it, optionally verifies batches, prints
    one JSON line."""
    a = argparse.Namespace(**spec["args"])
    batch, train, fanout = spec["batch"], np.fromfile(spec["train_file"], dtype=np.int32), spec["fanout"]
    import ipc_service
    torch.cuda.set_device(0)
    ipc_service.initialize()
    consume_lib, acc, c_p = None, None, None
    if a.consume:       # a consumer that READS every batch: one launch per get_next over the rows and the two outermost COO arrays
        import ctypes
        from legion_amd import lib as _lib
        consume_lib, c_p = _lib.load(), ctypes.c_void_p
        acc = torch.zeros(1, dtype=torch.float64, device="cuda:0")
        torch.cuda.synchronize()
        hp = torch.cuda.Stream(priority=-1)           # the consumer's one kernel per batch must not queue behind the server's saturating gathers
        stream = c_p(hp.cuda_stream)
    tr, va, te = ipc_service.get_steps()
    edges, t0, t1, n_timed, verified = 0, None, None, 0, 0
    t_start = time.time()
    n_all = (tr + va) * a.epochs + te                # the reference's schedule (ipc_service.cu:130-132)
    last_timed = (tr + va) * (a.epochs - 1) + tr - 1     # last training batch of the last epoch
    for i in range(n_all):
        if a.watchdog and time.time() - t_start > a.watchdog:
            raise RuntimeError(f"watchdog: stuck or too slow at batch {i} of {n_all}")
        out = ipc_service.get_next(a.dim)
        if i <= last_timed:
            if i == 5:
                torch.cuda.synchronize(); t0 = time.perf_counter(); edges = 0; n_timed = 0
            edges += int(out[3].numel())          # outermost block = every edge of the batch
            n_timed += 1
        if consume_lib is not None:
            f, s_, d_ = out[1], out[3], out[4]
            consume_lib.legion_consume_batch(stream, c_p(f.data_ptr()), f.numel(), c_p(s_.data_ptr()), c_p(d_.data_ptr()), s_.numel(), c_p(acc.data_ptr()))
            hp.synchronize()      # a batch is released (synchronize() below) only when its reads have completed
        if a.verify_every and i % a.verify_every == 0:
            ids, fts = out[0], out[1]
            n = int(ids.numel())
            assert n > 0 and tuple(fts.shape) == (n, a.dim)
            if not a.no_features_file:
                assert synth.feature_check_device(fts.contiguous(), ids.contiguous(), a.dim, 7) == 0, f"batch {i}: rows differ"
            assert int(torch.unique(ids).numel()) == n, f"batch {i}: duplicate ids"
            sizes = ipc_service.get_block_size()
            assert sizes[0] == n and int(out[3].max()) < n and int(out[4].max()) < sizes[1], f"batch {i}: edge endpoints"
            seeds_expected = torch.from_numpy(train[i * batch:(i + 1) * batch]).cuda() if (i < tr and a.epochs == 1) else None
            if seeds_expected is not None:
                assert bool((ids[:batch] == seeds_expected).all()), f"batch {i}: not the seeds of training batch {i}"
            verified += 1
        del out
        ipc_service.synchronize()
        if i == last_timed:
            torch.cuda.synchronize(); t1 = time.perf_counter()
    ipc_service.finalize()
    dt = t1 - t0
    print(json.dumps({"batches_per_sec": n_timed / dt, "edges_per_sec": edges / dt, "timed_batches": n_timed,
                      "ms_per_batch": dt / n_timed * 1e3, "verified_batches": verified,
                      "consumed_checksum": None if acc is None else float(acc.item())}), flush=True)


def proc_cpu_seconds(pid):
    """user + system CPU time of a live process, all threads (None when it is gone)"""
    try:
        f = open(f"/proc/{pid}/stat").read()
        rest = f[f.rindex(")") + 2:].split()
        return (int(rest[11]) + int(rest[12])) / os.sysconf("SC_CLK_TCK")
    except (OSError, ValueError, IndexError):
        return None


def thread_cpu_seconds(pid, detail=None):
    """{thread name: user + system CPU seconds} of a live process (threads of one name summed); `detail`, when given, gets
    {name: [user seconds, system seconds, voluntary context switches]}"""
    out = {}
    try:
        for tid in os.listdir(f"/proc/{pid}/task"):
            f = open(f"/proc/{pid}/task/{tid}/stat").read()
            name = f[f.index("(") + 1:f.rindex(")")]
            rest = f[f.rindex(")") + 2:].split()
            if not name.startswith("lg-"):         # (the runtime's helper threads carry the process's name: tell them apart)
                name = f"{name}:{'main' if int(tid) == int(pid) else int(tid) - int(pid)}"
            tck = os.sysconf("SC_CLK_TCK")
            out[name] = out.get(name, 0.0) + (int(rest[11]) + int(rest[12])) / tck
            if detail is not None:
                vol = 0
                for ln in open(f"/proc/{pid}/task/{tid}/status"):
                    if ln.startswith("voluntary_ctxt_switches"):
                        vol = int(ln.split()[1])
                d = detail.setdefault(name, [0.0, 0.0, 0, ""])
                d[0] += int(rest[11]) / tck
                d[1] += int(rest[12]) / tck
                d[2] += vol
                try:        # what the thread is doing right now: "running" or the number of the system call it is in (+ wchan)
                    sc = open(f"/proc/{pid}/task/{tid}/syscall").read().split()
                    d[3] = (sc[0] if sc else "?") + "/" + open(f"/proc/{pid}/task/{tid}/wchan").read().strip()
                except OSError:
                    d[3] = "?"
    except (OSError, ValueError, IndexError):
        pass
    return out


def run_one(a, ds, tmp, batch, n_train, train, fanout, N, E, mode=""):
    work = os.path.join(tmp, f"run_b{batch}{mode}")
    os.makedirs(work)
    open(os.path.join(work, "meta_config"), "w").write("{} {} {} {} {} {} {} {} {} {}".format(
        ds, batch, N, E, a.dim, n_train, batch, batch, a.cache_memory, a.epochs))
    ns = f"_b{os.getpid()}"
    os.environ["LEGION_IPC_NAMESPACE"] = ns
    if a.handover:
        os.environ["LEGION_RUNNER_HANDOVER"] = a.handover
    if a.no_views:
        os.environ["LEGION_NO_DIRECT_VIEWS"] = "1"

    def handover_of():
        for line in open(os.path.join(work, "server.log")):
            if "hand-over by" in line:
                return line.strip().split(": ", 1)[1]
        return "one gather launch per batch straight into the pipe slot" if os.environ.get("LEGION_RUNNER_HANDOVER") == "gather" else "?"
    workload = (f"N={N}, E={E} (synth.csr_device_large)" if a.nodes > 0 else f"RMAT-{a.scale} EF{a.edge_factor}") + f", D={a.dim}, batch {batch}, fanout {fanout}, train mode, 1 GPU" + \
               (", zero-filled feature table (no `features` file)" if a.no_features_file else "")
    log = open(os.path.join(work, "server.log"), "w")
    prefix = []
    if a.profile_server:
        prefix = ["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", os.path.join(a.profile_server, f"{mode or 'run'}_b{batch}"), "--"]
    server = subprocess.Popen(prefix + [os.path.join(ROOT, "legion_amd", "bin", "sampling_server"), "1", "0"] + [str(f) for f in fanout],
                              cwd=work, env=dict(os.environ, TMPDIR="/tmp"), stdout=log, stderr=subprocess.STDOUT)
    try:
        deadline = time.time() + 600
        while "System is ready for serving" not in open(os.path.join(work, "server.log")).read():
            assert server.poll() is None, open(os.path.join(work, "server.log")).read()
            assert time.time() < deadline
            time.sleep(0.2)
        if a.consumer == "native":
            exe = os.path.join(tmp, "boundary_consumer")
            subprocess.check_call(["gcc", "-O2", os.path.join(ROOT, "tools", "boundary_consumer.c"), "-o", exe, "-lrt", "-lpthread"])
            out = subprocess.check_output([exe, ns, "0", str(len(fanout)), "5", str(a.epochs), "0" if a.no_views else "1"]).decode().strip().splitlines()[-1]
            server.wait(timeout=120)
            log.flush()
            for line in open(os.path.join(work, "server.log")):
                if line.startswith("runner "):
                    print(line.strip(), file=sys.stderr)
            res = json.loads(out)
            res.update({"path": "sampling_server binary -> shm/semaphores -> protocol-only consumer (counters from the server's host-visible mirror)",
                        "workload": workload, "batch": batch, "handover": handover_of(), "mode": mode, "epochs": a.epochs})
            print(json.dumps(res), flush=True)
            return
        spec = {"args": {k: getattr(a, k) for k in ("dim", "epochs", "watchdog", "verify_every", "no_features_file", "consume")},
                "batch": batch, "fanout": fanout, "train_file": ds + "trainingset"}
        cpu0, w0 = proc_cpu_seconds(server.pid), time.time()
        child_p = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--child-consumer", json.dumps(spec)], env=dict(os.environ),
                                   stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        # the server's CPU time by thread WHILE it serves (its runner threads end with the schedule): snapshots every 0.2 s, the
        # window = first to last snapshot that shows a runner thread
        snaps = []
        seen = {}

        def watch():
            while child_p.poll() is None:
                det = {}
                t = thread_cpu_seconds(server.pid, det)
                if any(n.startswith("lg-runner") for n in t):
                    snaps.append((time.time(), t, det))
                    for n, d in det.items():
                        seen.setdefault(n, {})
                        seen[n][d[3]] = seen[n].get(d[3], 0) + 1
                time.sleep(0.2)
        import threading
        wt = threading.Thread(target=watch, daemon=True)
        wt.start()
        try:
            c_out, c_err = child_p.communicate(timeout=(a.watchdog or 3000) + 120)
        except subprocess.TimeoutExpired:
            child_p.kill()
            c_out, c_err = child_p.communicate()
        wt.join(timeout=2)
        child = argparse.Namespace(returncode=child_p.returncode, stdout=c_out, stderr=c_err)
        cpu1, w1 = proc_cpu_seconds(server.pid), time.time()
        by_thread, thread_detail = [], {}
        if len(snaps) >= 2:
            (ta, tha, da), (tb, thb, db) = snaps[0], snaps[-1]
            by_thread = sorted(((n, (thb.get(n, 0.0) - tha.get(n, 0.0)) / max(tb - ta, 1e-9)) for n in thb), key=lambda kv: -kv[1])[:6]
            zero = [0.0, 0.0, 0, ""]
            thread_detail = {n: {"user": round((db[n][0] - da.get(n, zero)[0]) / max(tb - ta, 1e-9), 3),
                                 "system": round((db[n][1] - da.get(n, zero)[1]) / max(tb - ta, 1e-9), 3),
                                 "voluntary_switches_per_s": round((db[n][2] - da.get(n, zero)[2]) / max(tb - ta, 1e-9)),
                                 "seen_in": dict(sorted(seen.get(n, {}).items(), key=lambda kv: -kv[1])[:4])}
                             for n, _ in by_thread if n in db}
        cl = [ln for ln in child.stdout.splitlines() if ln.startswith("{")]
        if child.returncode != 0 or not cl:
            raise RuntimeError(f"consumer failed (rc {child.returncode}): {child.stdout[-800:]} {child.stderr[-2500:]}")
        r = json.loads(cl[-1])
        server.wait(timeout=120)
        print(json.dumps({"path": "sampling_server binary -> shm/semaphores/IPC handles -> ipc_service consumer",
                          "workload": workload, "batch": batch, "handover": handover_of(), "mode": mode, "epochs": a.epochs,
                          "batches_per_sec": r["batches_per_sec"], "edges_per_sec": r["edges_per_sec"], "timed_batches": r["timed_batches"],
                          "ms_per_batch": r["ms_per_batch"], "verified_batches": r["verified_batches"],
                          "consumer_reads_every_batch": bool(a.consume), "consumed_checksum": r.get("consumed_checksum"),
                          "server_cpu_cores": None if cpu0 is None or cpu1 is None else (cpu1 - cpu0) / max(w1 - w0, 1e-9),
                          "server_cpu_cores_by_thread": {n: round(v, 3) for n, v in by_thread if v > 0.005},
                          "server_threads": thread_detail,
                          "server_cpu_cores_note": "user + system CPU seconds of the server process over the consumer's lifetime (incl. its "
                                                   "start-up, during which the server waits) / wall seconds"}), flush=True)
    finally:
        if server.poll() is None:
            server.kill()
        log.close()
        if a.watchdog:
            print("---- server log tail ----\n" + open(os.path.join(work, "server.log")).read()[-1500:], file=sys.stderr)
        for name in os.listdir("/dev/shm"):
            if name.endswith(ns):
                os.unlink(os.path.join("/dev/shm", name))


if __name__ == "__main__":
    if len(sys.argv) == 3 and sys.argv[1] == "--child-consumer":
        consume(json.loads(sys.argv[2]))
    else:
        main()
