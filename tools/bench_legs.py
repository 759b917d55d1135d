"""The legs of bench.py that are not the hot path's own timed region: argument parsing, the one-JSON-line guard, the two-phase
bulk pipeline wrapper, the drop-in boundary leg (server binary <-> ipc_service consumer), the in-run PMC traffic measurement, and
the pieces of a leg that bench.py's run_leg strings together (hotness collective, counting pass, timed regions, profiled pass,
cold-row regather, report assembly).  bench.py stays the entry point; nothing here touches oracle/ (bench.py's cpu_baseline does)."""
import argparse
import ctypes
import json
import os
import signal
import subprocess
import sys
import threading
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md: 8 TB/s spec
INFINITY_CACHE_BYTES = 256 << 20   # MI355X_MICROARCH.md: 256 MiB memory-side cache in front of the HBM
HBM_ACHIEVABLE_GBPS = 6290.0       # MI355X_MICROARCH.md: what a float4 streaming copy measures (79 % of the spec peak)


class OneLine:
    """The ONE JSON line of rank 0, written exactly once.  The legs behind the headline leg (N > 1: striped caches over peer pointers,
    never run on more than one physical GPU before the driver's SCALE run; N = 1: cold regather, other shapes, boundary, CPU baseline,
    traffic children) run with the guard ARMED: whatever ends this process while they run -- a Python error (caught by the caller), a
    native exit() of the library (a C atexit handler in tools/exit_line.c holding the serialised line), SIGTERM from the launcher
    (wake-up fd + watcher thread: works while the main thread is blocked inside a HIP or RCCL call), or nothing at all for
    `deadline_s` seconds (a hang) -- the headline leg's line, already measured, still goes out, with a note on what happened."""

    def __init__(self, fd):
        self.fd, self.line, self.done, self.lock = fd, None, False, threading.Lock()
        self._hook = None                   # the ctypes fallback (no C compiler at hand): then the process must leave through os._exit
        self._native = None                 # tools/_build/libexit_line.so
        self.needs_hard_exit = False

    def _serialise(self, obj):
        text = None
        for _ in range(5):          # (the fallback may serialise the object while the main thread adds a leg to it)
            try:
                text = json.dumps(obj)
                break
            except RuntimeError:
                time.sleep(0.01)
        if text is None:
            text = json.dumps({k: v for k, v in list(obj.items()) if k not in ("striped", "striped_replica", "striped_bulk", "other_shapes")})
        return text

    def emit(self, obj):
        with self.lock:
            if self.done:
                return
            self.done = True
        if self._native is not None:
            self._native.exit_line_clear()
        os.write(self.fd, (self._serialise(obj) + "\n").encode())

    def refresh(self):
        """After a leg added its fields to the armed object: the line the C exit handler holds follows."""
        if self._native is not None and self.line is not None and not self.done:
            note = dict(self.line)
            note["extra_legs_error"] = "the process exited inside a leg behind the headline leg (native exit)"
            text = (self._serialise(note) + "\n").encode()
            self._native.exit_line_set(self.fd, text, len(text))

    def _load_native(self):
        src = os.path.join(ROOT, "tools", "exit_line.c")
        lib = os.path.join(ROOT, "tools", "_build", "libexit_line.so")
        try:
            if not os.path.exists(lib) or os.path.getmtime(lib) < os.path.getmtime(src):
                os.makedirs(os.path.dirname(lib), exist_ok=True)
                tmp = lib + ".%d" % os.getpid()
                subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", src, "-o", tmp], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                os.replace(tmp, lib)
            L = ctypes.CDLL(lib)
            L.exit_line_set.argtypes = [ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t]
            L.exit_line_set.restype = None
            L.exit_line_clear.restype = None
            return L
        except Exception:       # noqa: BLE001 -- no compiler, read-only tree ...: the ctypes hook below stands in
            return None

    def arm(self, headline_obj, deadline_s=900):
        """From here on a dying -- or, after `deadline_s`, a hung -- process still prints `headline_obj` -- the object itself, not a
        copy: whatever extra legs have been added to it by then go out with it (call refresh() after each)."""
        self.line = headline_obj

        def fallback(why):
            if self.line is not None and not self.done:
                self.line["extra_legs_error"] = why
                self.emit(self.line)

        self._native = self._load_native()
        if self._native is not None:
            self.refresh()
        else:
            # a Python callable registered with libc (glibc exports __cxa_atexit): it would also be called after the interpreter is
            # gone on a normal exit, so the caller must leave through os._exit() once it has printed the line
            self._hook = ctypes.CFUNCTYPE(None, ctypes.c_void_p)(lambda _: fallback("the process exited inside an extra leg (native exit)"))
            getattr(ctypes.CDLL(None), "__cxa_atexit")(self._hook, None, None)
            self.needs_hard_exit = True
        rfd, wfd = os.pipe()
        os.set_blocking(wfd, False)
        signal.set_wakeup_fd(wfd, warn_on_full_buffer=False)
        signal.signal(signal.SIGTERM, lambda *_: None)       # (the C-level handler writes the signal number to wfd)

        def watch():
            import select
            got, _, _ = select.select([rfd], [], [], deadline_s)
            fallback("SIGTERM while a leg behind the headline leg was running (another rank failed? a time limit?)" if got else
                     "the legs behind the headline leg did not finish within %d s (hung peer load, collective or child process?)" % deadline_s)
            os._exit(1)

        threading.Thread(target=watch, daemon=True).start()
        return fallback


class BulkPipe:
    """engine.Pipeline's submit / wait / run_range interface over the two-phase bulk protocol (pipeline.hip): phase A on every
    rank -> barrier -> phase B on every rank -> barrier.  No hipGraphs, no per-gather HIP events."""

    def __init__(self, pipe, use_dist):
        self.p, self.pools, self.use_dist = pipe, pipe.pools, use_dist
        self.group_size = pipe.group_size
        pipe.bulk_enable()
        self.count_rows, self.rows_listed = False, 0
        self.reset_clocks()

    def reset_clocks(self):
        self.t_a = self.t_b = self.t_bar = 0.0
        self.groups = 0

    def submit(self, counter0, mode=0, n_active=None):
        t0 = time.perf_counter()
        slot = self.p.bulk_phase_a(counter0, mode, n_active)
        t1 = time.perf_counter()
        if self.use_dist:
            dist.barrier()
        t2 = time.perf_counter()
        if self.count_rows:
            self.rows_listed += self.p.bulk_listed(slot)
            t2 = time.perf_counter()
        self.p.bulk_phase_b(slot)
        t3 = time.perf_counter()
        if self.use_dist:
            dist.barrier()
        t4 = time.perf_counter()
        self.t_a += t1 - t0
        self.t_b += t3 - t2
        self.t_bar += (t2 - t1) + (t4 - t3)
        self.groups += 1
        return slot

    def run_range(self, first, count, mode=0, wrap=None):
        k, last = 0, None
        while k < count:
            b = (first + k) % wrap if wrap else first + k
            n = min(self.group_size, count - k, (wrap - b) if wrap else count)
            last = (self.submit(b, mode, n), b, n)
            k += n
        return last

    def wait(self, slot=-1):
        pass                                        # both phases synchronise their stream

    def profile_begin(self):
        pass

    def profile_end(self):
        pass

    def profile_read(self):
        return {}

    def close(self):
        self.p.close()


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=32, help="timed steps; a step = one launch group of --group mini-batches")
    ap.add_argument("--warmup", type=int, default=8, help="untimed warm-up steps (launch groups)")
    ap.add_argument("--min-seconds", type=float, default=1.5,
                    help="repeat the K-step timed region until this much time has been measured (median reported)")
    ap.add_argument("--max-repeats", type=int, default=400)
    ap.add_argument("--placement", type=str, default="hbm", choices=["hbm", "pinned"],
                    help="pinned: full CSR and full feature table in mapped pinned host memory (the reference's only tier; "
                         "BASELINE configs[2]): cache hits come from HBM, misses are read in place over PCIe")
    ap.add_argument("--scramble", action="store_true",
                    help="Graph500-style label scrambling of the RMAT vertices (hubs no longer sit at the low ids)")
    ap.add_argument("--scale", type=int, default=26)
    ap.add_argument("--nodes", type=int, default=0,
                    help="with --edges: a skewed synthetic graph of exactly this many vertices and edges instead of RMAT-<scale> "
                         "(synth.csr_device_large: any vertex count, more than 2^32 edges; the reference's real data-set sizes, "
                         "legion_server.py:41-88, e.g. uk-union --nodes 133633040 --edges 5507679822)")
    ap.add_argument("--edges", type=int, default=0)
    ap.add_argument("--edge-factor", type=int, default=16)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--fanout", type=str, default="25,10")
    ap.add_argument("--cache-memory", type=int, default=8 << 30, help="bytes per GPU fed to the cost model")
    ap.add_argument("--presc-steps", type=int, default=512, help="PreSC batches per GPU (bounded epoch)")
    ap.add_argument("--cpu-seconds", type=float, default=16.0,
                    help="target time of EACH CPU-baseline leg (Legion-semantics port, DGL-semantics port); 0 disables")
    ap.add_argument("--group", type=int, default=0,
                    help="mini-batches served by every launch (lanes of a group); 0 = 524288 // batch rounded down to a power of "
                         "two, at most 512, halved while the lanes in flight would not fit 0.7 of the free HBM")
    ap.add_argument("--slots", type=int, default=2, help="groups in flight per GPU")
    ap.add_argument("--lanes", dest="lane_arena", default=True, type=lambda v: {"arena": True, "plain-arena": "plain", "separate": False}[v],
                    help="where the lanes' trainer-visible arrays live: arena (default: one arena of shuffled physical chunks), plain-arena (one plain "
                         "allocation: what the server's hand-over needs), separate (an allocation per array and lane)")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--overlap", action="store_true", help="let kernels of different slots share the GPU")
    ap.add_argument("--no-weave", action="store_true",
                    help="everything of a group on ONE stream (default: the next group's head -- seeds + every hop but the last, small "
                         "latency-bound kernels -- runs on a second stream under the current group's heavy kernels, pipeline.hip)")
    ap.add_argument("--capacity", type=str, default="",
                    help="NODE,EDGE: cache capacities per GPU set by hand after the cost model has run (its choice is logged): e.g. a "
                         "topology cache of the EDGE hottest vertices' adjacency beside a pinned-host CSR (SURVEY section 8 N1)")
    ap.add_argument("--hybrid", type=str, default="",
                    help="CPU_ROWS,GPU_ROWS: the hybrid CPU-cache / GPU-cache tier (UnifiedCache::HybridInit, SS/cache/cache.cu:614-670; what the "
                         "server builds in disk mode) instead of CandidateSelection + CostModel + FillUp: the GPU_ROWS hottest rows of this GPU's own "
                         "order in an HBM cache, the next CPU_ROWS in a mapped pinned host cache (read over PCIe), the rest from the full table")
    ap.add_argument("--no-cache", action="store_true",
                    help="experiment: no feature/topology cache at all (no FillUp): every row comes from the full table and the "
                         "gather makes no node_map lookup -- what the lookup's 128-byte line per row costs the gather")
    ap.add_argument("--gather-rows", type=int, default=0, help="experiment: rows per gather workgroup (LegionTuning.gather_rows_per_wg)")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-overlap-leg", action="store_true", help=argparse.SUPPRESS)      # (accepted, ignored: the leg it skipped was removed in round 5)
    ap.add_argument("--no-boundary", action="store_true",
                    help="skip the drop-in boundary leg (sampling_server binary -> shm/semaphores/IPC handles -> ipc_service "
                         "consumer on an RMAT-22 data set written in the reference's file formats; N = 1 only, ~15 s)")
    ap.add_argument("--no-traffic-leg", action="store_true",
                    help="skip the measurement of roofline.traffic in this run (two fresh child processes of this command under "
                         "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, N = 1 only; also skipped with --no-boundary)")
    ap.add_argument("--traffic-deadline", type=int, default=300, help="seconds each of those child processes may take")
    ap.add_argument("--no-other-shapes", action="store_true",
                    help="skip the `other_shapes` legs (N = 1, the default shape only: B = 8000 at [25,10] and at [15,10,5] on the same tables, "
                         "--other-shapes-steps timed steps each, ~25 s each)")
    ap.add_argument("--other-shapes-steps", type=int, default=5)
    ap.add_argument("--cold-leg", action="store_true", help="run roofline.cold / roofline.alone even with --no-boundary (they belong to the full default run)")
    ap.add_argument("--no-cold-leg", action="store_true",
                    help="skip roofline.cold / roofline.alone (the last hop's gather launched again alone, and over rows that never repeat)")
    ap.add_argument("--boundary-modes", type=str, default="views,slab",
                    help="boundary leg: hand-overs to measure, one server run per mode and batch size (tools/server_throughput.py --modes: "
                         "views | slab | gather | copy); tools/profile_round.sh adds copy")
    ap.add_argument("--boundary-batches", type=int, default=40000,
                    help="boundary leg: timed batches per server run at least (the server runs as many epochs as that takes)")
    ap.add_argument("--measured-counters", action="store_true", help="same as --link-counters computed")
    ap.add_argument("--link-counters", type=str, default="v2", choices=["v2", "computed", "smi"],
                    help="what feeds CostModel's PCIe transaction counters: v2 = {0,0} as the reference's v2 does; computed = the "
                         "64-byte topology transactions the sampler counted during PreSC; smi = what the PCIe link really carried "
                         "during PreSC, from the driver's cumulative gpu_metrics counter (the paper's Intel-PCM reading)")
    ap.add_argument("--stripe", action="store_true",
                    help="N > 1: one clique of N GPUs, feature/topology caches striped over the ranks and read "
                         "through peer pointers over xGMI (default: every GPU caches for itself, no peer traffic)")
    ap.add_argument("--replica-memory", type=int, default=0,
                    help="with --stripe: bytes per GPU for a private copy of the clique's hottest rows (hits below that hotness "
                         "rank are read from local HBM instead of a peer over xGMI; lookup results unchanged)")
    ap.add_argument("--no-striped-leg", action="store_true",
                    help="N > 1 without --stripe: skip the two extra timed legs with the caches striped over one clique of N "
                         "(plain, and with a hot-row replica of --striped-replica-memory bytes)")
    ap.add_argument("--no-bulk-leg", action="store_true",
                    help="N > 1: skip the `striped_bulk` leg (striped caches, remote rows pushed by their owners: peer_gather = bulk)")
    ap.add_argument("--striped-replica-memory", type=int, default=4 << 30,
                    help="bytes per GPU of the hot-row replica in the `striped_replica` leg")
    ap.add_argument("--backend", type=str, default="nccl", help="torch.distributed backend (nccl = RCCL)")
    ap.add_argument("--no-product-collective", action="store_true",
                    help="hotness all-reduce through torch.distributed instead of the library's own RCCL call")
    ap.add_argument("--collective-deadline", type=int, default=90,
                    help="seconds the library's communicator may take to form before the run falls back to torch.distributed")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed and run the collectives even at N = 1 (exercises the RCCL calls on a 1-GPU box)")
    ap.add_argument("--extra-legs-deadline", type=int, default=330,
                    help="N > 1: seconds the extra (striped) legs may take together before rank 0 prints the headline line alone and exits "
                         "(budget of an N = 8 run against a 600 s limit: headline leg <= ~150 s even when the library's communicator never forms "
                         "-- set-up ~60 s + --collective-deadline 90 s -- plus these 330 s)")
    ap.add_argument("--post-legs-deadline", type=int, default=420,
                    help="N = 1: seconds the legs behind the headline leg (cold regather, other shapes, boundary, CPU baseline, traffic children; "
                         "~170 s when nothing hangs) may take together before the headline line goes out alone")
    ap.add_argument("--fail-extra-leg", type=str, default="", choices=["", "raise", "exit", "sigterm", "hang"],
                    help="testing: make rank 0 fail this way when the first extra leg starts (the headline line must still go out)")
    ap.add_argument("--force-device", type=int, default=-1,
                    help="put every rank on this GPU (testing the N > 1 code path on a 1-GPU box, with --backend gloo)")
    return ap.parse_args()


def boundary_leg(args, fanout):
    """The same kernels behind the reference's own server <-> trainer protocol (tools/server_throughput.py): the
    `sampling_server` binary serving a Python `ipc_service` consumer one mini-batch per semaphore hand-off into one of two
    pipe slots -- on the bench's own graph (RMAT-26) at the bench's batch size AND at Legion's default B = 8000, where the
    per-batch hand-over latency no longer hides the GPU.  Reported beside the headline, never as `value`.  The data set is
    written to /tmp in the reference's file formats (CSR 4.8 GB at RMAT-26; no `features` file: the server serves a zero-filled
    table of the same shape, as v2 of the reference does, storage_management.cu:162); without room there the leg falls back
    to RMAT-22 and says so."""
    import shutil
    import subprocess
    scale = args.scale
    need = (1 << scale) * (8 + 4 * args.edge_factor + 4) + (2 << 30)
    note = None
    try:
        free = shutil.disk_usage("/tmp").free
    except OSError:
        free = 0
    if free < need:
        note = f"/tmp has {free >> 20} MiB free, the RMAT-{scale} data set needs {need >> 20}: boundary leg run at RMAT-22 instead"
        scale = min(scale, 22)
    batches = [args.batch] + ([8000] if args.batch != 8000 else [])
    # how a batch reaches the trainer end (LegionTuning.runner_handover, server.hip): `views` -- whole launch groups into the
    # server's lane arena, this build's ipc_service takes every batch as views of its lane: what a user of legion_graphsage.py
    # gets; `slab` -- the same server with a trainer end that opens only the reference's slab: one gather launch per batch
    # straight into the pipe slot (round 3's path; the COMPATIBILITY path, not the fast one)
    cmd = [sys.executable, os.path.join(ROOT, "tools", "server_throughput.py"), "--scale", str(scale), "--edge-factor", str(args.edge_factor),
           "--batch", ",".join(str(b) for b in batches), "--dim", str(args.dim), "--fanout", ",".join(str(f) for f in fanout),
           "--train-batches", str(max(64, min(3072, (3 << 20) // args.batch))), "--no-features-file", "--cache-memory", str(args.cache_memory),
           "--modes", args.boundary_modes, "--min-timed-batches", str(args.boundary_batches), "--watchdog", "800"]
    try:
        res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1200)
        lines = [json.loads(ln) for ln in res.stdout.splitlines() if ln.startswith("{")]
        if not lines:
            raise RuntimeError(res.stderr[-300:])
        def leg_of(r):
            return {"mode": r.get("mode"), "batch": r["batch"], "batches_per_sec": r["batches_per_sec"], "edges_per_sec": r["edges_per_sec"],
                    "handover": r.get("handover"), "path": r["path"], "workload": r["workload"], "ms_per_batch": r["ms_per_batch"],
                    "timed_batches": r["timed_batches"], "epochs": r.get("epochs"), "server_cpu_cores": r.get("server_cpu_cores")}
        legs = [leg_of(r) for r in lines]
        # ... and once more with a consumer that READS every batch it is handed (one launch per get_next over the rows and the
        # outermost COO pair, completed before the batch is released): the protocol-only figure above is a rate of hand-overs
        # nobody looks at; this one shares the HBM with the server's gathers like a training loop's first layer would
        consuming = None
        try:
            i = cmd.index("--modes")
            cmd2 = cmd[:i] + ["--modes", "views"] + cmd[i + 2:]
            i = cmd2.index("--min-timed-batches")
            cmd2[i + 1] = str(max(2000, args.boundary_batches // 4))
            res2 = subprocess.run(cmd2 + ["--consume"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
            l2 = [json.loads(ln) for ln in res2.stdout.splitlines() if ln.startswith("{")]
            if not l2:
                raise RuntimeError(res2.stderr[-300:])
            consuming = {"by_batch_size": [leg_of(r) for r in l2],
                         "note": "views hand-over with a trainer end that reads every batch: legion_consume_batch (one launch: every float of "
                                 "the rows + the outermost COO pair) and a stream synchronise before each synchronize(), as the protocol "
                                 "demands of a consumer that reads in place; the consumer's reads share the HBM with the server's gathers "
                                 "(+ rows x D x 4 bytes of traffic per batch), and at B = 1024 its per-batch launch + synchronise "
                                 "(~10 us of host time) exceeds the 6.3 us the server needs per batch"}
        except Exception as e:
            consuming = {"error": repr(e)[:300]}
        first = legs[0]
        out = {"boundary_batches_per_sec": first["batches_per_sec"], "boundary_edges_per_sec": first["edges_per_sec"],
               "boundary": {"path": first["path"], "workload": first["workload"], "ms_per_batch": first["ms_per_batch"],
                            "timed_batches": first["timed_batches"], "handover": first["handover"],
                            "what_it_measures": "the hand-over protocol with a consumer that never reads a row (zero-filled feature table of the "
                                                "right shape): the rate at which batches CAN be taken; `consuming_trainer` reads them",
                            "by_batch_size": [l for l in legs if l["mode"] == "views"],
                            "consuming_trainer": consuming,
                            "slab_only_trainer": {"note": "the COMPATIBILITY path: a trainer end that opens only the reference's slab (a build of "
                                                          "TB/ipc_cuda_kernel.cu; no views of the lane arena) gets every batch gathered into the "
                                                          "pipe slot by one launch, two slots in flight -- bound by the launch -> completion -> "
                                                          "semaphore round trip per batch, not by the GPU",
                                                  "by_batch_size": [l for l in legs if l["mode"] != "views"]}}}
        if note:
            out["boundary"]["note"] = note
        return out
    except Exception as e:            # the headline must not depend on this leg
        return {"boundary_batches_per_sec": None, "boundary": {"error": repr(e)[:300]}}


def measured_traffic(args, roof, G):
    """roofline.traffic, MEASURED in this run: two fresh child processes of this very command -- started as children, never an exec of
    this process -- under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `... WRITE_SIZE` (separate passes, counters beside the
    kernel trace only, the interpreter binary directly behind `--`: MI355X_MICROARCH.md, HBM), two timed steps each on the same
    workload and group size; folded as tools/pmc_summary.py folds the committed profile (both counters KiB; FETCH_SIZE doubled:
    gfx950 tallies the 128-byte requests of a 16-byte-per-lane stream at 64 B).  Sets roof["traffic"] (bytes per launch; it stays
    null when a child fails or hangs), traffic_over_algorithmic and traffic_source; the committed profile's figure, rescaled by this
    run's rows, is beside it as traffic_committed."""
    import csv
    import glob
    import shutil
    import tempfile
    prof = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if prof is None:
        roof["traffic_note"] = "rocprofv3 not found on this box"
        return
    t0 = time.time()
    shape = ["--scale", str(args.scale), "--edge-factor", str(args.edge_factor), "--dim", str(args.dim), "--batch", str(args.batch),
             "--fanout", args.fanout, "--group", str(G), "--cache-memory", str(args.cache_memory)]
    if args.nodes > 0:
        shape += ["--nodes", str(args.nodes), "--edges", str(args.edges)]
    if args.scramble:
        shape += ["--scramble"]
    child = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--presc-steps", "64", "--cpu-seconds", "0",
             "--no-verify", "--no-boundary", "--no-other-shapes", "--no-cold-leg", "--min-seconds", "0.01"] + shape
    got, rows_child, note = {}, None, None
    tmp = tempfile.mkdtemp(prefix="legion_pmc_", dir="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            os.makedirs(d)
            cmd = [prof, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "--"] + child
            try:
                res = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                     stdin=subprocess.DEVNULL, text=True, timeout=args.traffic_deadline)
            except subprocess.TimeoutExpired:
                note = f"the {counter} pass did not finish within {args.traffic_deadline} s"
                break
            lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
            files = glob.glob(d + "/*/*counter_collection.csv") + glob.glob(d + "/*counter_collection.csv")
            if res.returncode != 0 or not lines or not files:
                note = f"the {counter} pass failed (rc {res.returncode}): {res.stderr[-300:]}"
                break
            rows_child = json.loads(lines[-1])["roofline"]["rows_per_launch"]
            sel = [r for r in csv.DictReader(open(files[0])) if r["Counter_Name"] == counter and "gather_kernel" in r["Kernel_Name"] and
                   r["Kernel_Name"][:r["Kernel_Name"].rfind("(")].rstrip().endswith("true>")]
            if not sel:
                note = f"no launch of the last-hop gather in the {counter} pass"
                break
            full = max(int(r["Grid_Size"]) for r in sel)                       # launches over a full group
            vals = [float(r["Counter_Value"]) for r in sel if int(r["Grid_Size"]) == full]
            got[counter] = (sum(vals) / len(vals), len(vals))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    if note is not None or len(got) != 2 or not rows_child:
        roof["traffic_note"] = note or "incomplete"
        return
    read_b, write_b = 2 * got["FETCH_SIZE"][0] * 1024, got["WRITE_SIZE"][0] * 1024
    alg = rows_child * roof["bytes_per_row"]
    roof["traffic"] = read_b + write_b
    roof["traffic_read_bytes"], roof["traffic_write_bytes"] = read_b, write_b
    roof["traffic_over_algorithmic"] = (read_b + write_b) / alg
    roof["traffic_source"] = "this run"
    roof["traffic_note"] = ("bytes per launch of the last hop's gather over a full group, averaged over %d / %d launches of two child "
                                     "processes of this command under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (%d rows per launch there; "
                                     "FETCH_SIZE doubled per the guide's gfx950 correction, both KiB); %.0f s" %
                                     (got["FETCH_SIZE"][1], got["WRITE_SIZE"][1], rows_child, time.time() - t0))




# ---- the pieces of one leg (bench.py: run_leg) ---------------------------------------------------------------------------------------
def hotness_collective(c, engine, cache, d, red_dev):
    """The only collective of the path: the all-reduce of the two uint64[N] hotness arrays.  The PRODUCT issues it
    (legion_amd/csrc/collective.hip: ncclAllReduce(ncclUint64, ncclSum) over its own communicator); torch.distributed only carries
    rank 0's 128-byte unique id to the other ranks.  A join or a call that fails or does not return within --collective-deadline
    seconds falls back to dist.all_reduce on the same arrays and the line says so -- a SCALE run must not be lost to the first
    meeting of this code with a second physical GPU."""
    args, world, rank, N = c.args, c.world, c.rank, c.N
    ones = torch.ones(1, dtype=torch.int64, device=red_dev)
    dist.all_reduce(ones)                                   # the world size as torch.distributed sees it
    torch.cuda.synchronize()
    issued_by, product_err, world_seen, ar_ms = None, None, 0, 0.0
    if args.backend == "nccl" and not args.no_product_collective and not getattr(c, "product_collective_broken", False):
        ids = [engine.collective_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        box = {}

        def product_call():
            try:
                if not engine.collective_init_rank(ids[0], world, rank, d):
                    box["err"] = "legion_collective_init_rank failed"
                    return
                box["joined"] = True
            except Exception as e:      # noqa: BLE001
                box["err"] = repr(e)[:200]

        th = threading.Thread(target=product_call, daemon=True)
        th.start()
        th.join(args.collective_deadline)
        ok_t = torch.tensor([1 if box.get("joined") else 0], dtype=torch.int64, device=red_dev)
        dist.all_reduce(ok_t, op=dist.ReduceOp.MIN)         # every rank takes the same path
        if int(ok_t.item()) == 1:
            dist.barrier()
            t0 = time.perf_counter()
            world_seen, _ = cache.allreduce_hotness(d)
            torch.cuda.synchronize()
            ar_ms = (time.perf_counter() - t0) * 1e3
            ok_t = torch.tensor([1 if world_seen == world else 0], dtype=torch.int64, device=red_dev)
            dist.all_reduce(ok_t, op=dist.ReduceOp.MIN)
            if int(ok_t.item()) == 1:
                issued_by = "liblegion_hip.so (collective.hip: ncclAllReduce, ncclUint64, ncclSum, in place, the library's own communicator)"
            else:
                raise RuntimeError("the product's hotness all-reduce ran on some ranks only: the counters are inconsistent")
        else:
            product_err = box.get("err", f"no join within {args.collective_deadline} s")
            c.product_collective_broken = True      # (a join that is still stuck holds the library's lock: later legs do not try again)
    if issued_by is None:
        dist.barrier()
        t0 = time.perf_counter()
        dist.all_reduce(cache.array("node_access_time", d))
        dist.all_reduce(cache.array("edge_access_time", d))
        torch.cuda.synchronize()
        ar_ms = (time.perf_counter() - t0) * 1e3
        world_seen = int(ones.item())
        issued_by = f"torch.distributed ({dist.get_backend()})" + (" -- the product's own call was not used: " + product_err if product_err else "")
    ar_t = torch.tensor([ar_ms], dtype=torch.float64, device=red_dev)
    dist.all_reduce(ar_t, op=dist.ReduceOp.MAX)
    return {"backend": "rccl" if "liblegion" in issued_by else dist.get_backend(), "issued_by": issued_by,
            "world_size_seen_by_all_reduce": int(world_seen),
            "hotness_all_reduce_ms": float(ar_t.item()), "hotness_all_reduce_bytes": 2 * N * 8,
            "hotness_all_reduce_GBps_algorithmic": 2 * N * 8 / max(float(ar_t.item()), 1e-6) / 1e6,
            "note": "two uint64[N] arrays (node and edge access counts), all-reduced in place once before CandidateSelection; "
                    "time = max over ranks, wall clock around both calls incl. synchronize"}


def counting_pass(c, synth, pipe, cache, d, node_map, feature_rows, headline, stripe, bulk):
    """Untimed, deterministic pass over exactly the timed batches: per batch its edges, rows per hop, slots per hop; over the first
    launch group the hit rate and how many of the last hop's rows are DISTINCT (roofline.unique_row_frac: a row gathered by several
    mini-batches of one launch may be served by the 256 MiB Infinity Cache the second time); the headline leg also verifies the
    size-independent parity properties on the first batch."""
    args, H, D, G, n_timed, wrap, dev = c.args, c.H, c.D, c.G, c.n_timed, c.wrap, c.dev
    first = c.n_warm
    r = type("Counted", (), {})()
    r.edges = np.zeros(n_timed, dtype=np.int64)
    r.rows = np.zeros((n_timed, H + 1), dtype=np.int64)
    r.hop_edges = np.zeros((n_timed, H), dtype=np.int64)
    r.hop_slots = np.zeros((n_timed, H), dtype=np.int64)
    r.hits = r.feat_hit_rows = r.feat_miss_rows = 0
    hybrid_cpu_cap = int(args.hybrid.split(",")[0]) if args.hybrid else None
    tiers = [0, 0, 0]
    last_hop_ids = []
    if stripe:
        cache.gather_stats3(d)                   # arms the row-source counters for this (untimed) pass only
    if bulk:
        pipe.count_rows = True
    for k in range(n_timed):
        if k % G == 0:
            slot = pipe.submit((first + k) % wrap if wrap else first + k)
            pipe.wait(slot)
        pl = pipe.pools[slot][k % G]
        nc = pl.buffer("node_counter").cpu().numpy()
        ec = pl.buffer("edge_counter").cpu().numpy()
        r.edges[k] = ec[9 + H]
        r.rows[k, 0] = nc[9]
        if node_map.numel() > 0 and (args.placement == "pinned" or k < G):
            slots = node_map[pl.buffer("sampled_ids")[:int(nc[9 + H])].long()]
            hm = slots >= 0
            r.feat_hit_rows += int(hm.sum())
            r.feat_miss_rows += int(hm.numel() - int(hm.sum()))
            if hybrid_cpu_cap is not None:       # hybrid tier: slots below cpu_cap are CPU-cache rows (cache_impl.cuh:224-231)
                in_cpu = int(((slots >= 0) & (slots < hybrid_cpu_cap)).sum())
                tiers[0] += in_cpu
                tiers[1] += int(hm.sum()) - in_cpu
                tiers[2] += int(hm.numel() - int(hm.sum()))
        if k < G and headline:
            last_hop_ids.append(pl.buffer("sampled_ids")[int(nc[9 + H - 1]):int(nc[9 + H])].clone())
        for h in range(H):
            r.rows[k, h + 1] = nc[9 + h + 1] - nc[9 + h]
            r.hop_edges[k, h] = ec[9 + h + 1] - ec[9 + h]
            r.hop_slots[k, h] = (nc[9] if h == 0 else ec[9 + h] - ec[9 + h - 1]) * c.fanout[h]
        if k == 0 and headline and not args.no_verify:
            # size-independent parity properties at full size: every gathered row is byte-identical to
            # the generator's value for its id; ids are unique; positions localise the edge endpoints
            n = int(nc[9 + H])
            assert n <= feature_rows
            ids = pl.buffer("sampled_ids")[:n]
            bad_words = synth.feature_check_device(pl.buffer("float_features")[:n].contiguous(), ids.contiguous(), D, 7)
            assert bad_words == 0, f"{bad_words} gathered words differ from the source rows"
            assert int(torch.unique(ids).numel()) == n, "duplicate node ids in the batch"
            e = int(ec[9 + H])
            src_g = pl.buffer("agg_src_ids")[:e].long()
            assert bool((ids.long()[pl.buffer("agg_src_off")[:e].long()] == src_g).all())
            r.hits = int((pl.buffer("cache_search_buffer")[:int(nc[1])] >= 0).sum())
    r.tiers = tuple(tiers) if hybrid_cpu_cap is not None else None
    r.unique = None
    if last_hop_ids:
        allids = torch.cat(last_hop_ids)
        uniq, cnt = torch.unique(allids, return_counts=True)
        r.unique = {"rows": int(allids.numel()), "unique_rows": int(uniq.numel()), "rows_gathered_once": int((cnt == 1).sum()),
                    "repeat_rows": int(allids.numel() - uniq.numel())}
        del allids, uniq, cnt, last_hop_ids
    if bulk:
        pipe.count_rows = False
        pipe.reset_clocks()
    r.source_rows = None
    if stripe:                                   # where this rank's gathers read the timed batches' hit rows from
        r.source_rows = cache.gather_stats3(d)
        cache.gather_stats_enable(False)         # counting costs an atomic per hit row: off before anything is timed
    return r


def timed_regions(c, engine, pipe, d, headline):
    """Warm-up, then the timed region: exactly K steps (K hipGraph replays of G batches each) between barrier + synchronize
    brackets.  The region is repeated (same batches: an epoch over the same seeds, replays are deterministic) until --min-seconds
    have been timed; every rank runs the same number of repeats, per repeat the MAX over ranks counts, the median repeat is reported."""
    args, use_dist, dev = c.args, c.use_dist, c.dev
    first, n_timed, wrap = c.n_warm, c.n_timed, c.wrap
    pipe.run_range(0, c.n_warm, wrap=wrap)
    pipe.wait()

    def timed_region():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        last = pipe.run_range(first, n_timed, wrap=wrap)
        pipe.wait()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        return time.perf_counter() - t0, last

    t = type("Timed", (), {})()
    min_seconds = args.min_seconds if headline else 0.5 * args.min_seconds
    time.sleep(0.005)
    t.lk0, t.t_lk0 = engine.link_counters_ex(d), time.perf_counter()
    el0, t.last_group = timed_region()
    reps_t = torch.tensor([max(1, min(args.max_repeats, int(np.ceil(min_seconds / max(el0, 1e-6)))))], dtype=torch.int64, device=dev)
    if use_dist:
        dist.all_reduce(reps_t, op=dist.ReduceOp.MAX)
    t.repeats = int(reps_t.item())
    region_s = [el0]
    for _ in range(t.repeats - 1):
        el, t.last_group = timed_region()
        region_s.append(el)
    time.sleep(0.005)
    t.lk1, t.t_lk1 = engine.link_counters_ex(d), time.perf_counter()
    t.own_region = float(np.median(np.asarray(region_s)))                     # this rank's own clock (brackets include the barriers)
    region_t = torch.tensor(region_s, dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(region_t, op=dist.ReduceOp.MAX)              # per repeat: the slowest rank
    t.region_s = region_t.cpu().numpy()
    t.elapsed_max = float(np.median(t.region_s))
    return t


def verify_last_group(c, synth, pipe, counted, last_group):
    """The last group of the timed region is still in its slot: its batches must be the ones the counting pass saw (replays are
    deterministic), and its last lane passes the full-size property checks."""
    H, D, n_timed = c.H, c.D, c.n_timed
    slot, k0, n_lanes = last_group
    for lane in range(n_lanes):
        pl = pipe.pools[slot][lane]
        nc = pl.buffer("node_counter").cpu().numpy()
        ec = pl.buffer("edge_counter").cpu().numpy()
        k = n_timed - n_lanes + lane          # the last group submitted holds the last n_lanes batches of the region
        assert ec[9 + H] == counted.edges[k] and nc[9 + H] - nc[9] == counted.rows[k, 1:].sum(), f"replayed batch {k0 + lane} differs"
    n = int(nc[9 + H])
    ids = pl.buffer("sampled_ids")[:n]
    assert synth.feature_check_device(pl.buffer("float_features")[:n].contiguous(), ids.contiguous(), D, 7) == 0
    assert int(torch.unique(ids).numel()) == n


def profiled_pass(c, pipe):
    """The same K batches once more with HIP events around every gather launch (recorded on the lane's own stream).  Eager launches:
    HIP cannot time events recorded by graph nodes.  -> ({op: (ms, launches)}, wall seconds of the pass)"""
    pipe.profile_begin()
    pipe.run_range(0, c.n_warm, wrap=c.wrap)
    pipe.wait()
    warm = pipe.profile_read()
    t1 = time.perf_counter()
    pipe.run_range(c.n_warm, c.n_timed, wrap=c.wrap)
    pipe.wait()
    elapsed_profiled = time.perf_counter() - t1
    prof = pipe.profile_read()
    pipe.profile_end()
    prof = {op: (ms - warm.get(op, (0.0, 0))[0], cnt - warm.get(op, (0.0, 0))[1]) for op, (ms, cnt) in prof.items()}
    err_bits = 0                              # LG_ERR_* bits a kernel raised for any lane (table full, feature rows, chain)
    for row in pipe.pools:
        for pl in row:
            err_bits |= pl.error()
    if err_bits:
        raise RuntimeError(f"a kernel raised error bits {err_bits:#x} during the run (legion_core.h LG_ERR_*)")
    return prof, elapsed_profiled


def cold_regather(c, pipe, node_map, bytes_per_row, repeats=7):
    """how much of the headline gather's rate is the part's memory-side cache.  The last hop's gather of one launch
    group is launched again ALONE (nothing beside it on the machine) over the lanes as they stand -- `alone` -- and then once more
    after the last hop's ids of every lane were replaced by ids of which none repeats anywhere in the launch (a slice of a random
    permutation of the vertices; the carried cache slots rewritten to match): `cold`.  Same kernel instance, same grid, same rows
    per lane, same hit / miss tiers (both in HBM); only the reuse is gone."""
    H, G, N, dev = c.H, c.G, c.N, c.dev
    slot = pipe.submit(c.n_warm)
    pipe.wait(slot)
    spans, total = [], 0
    for lane in range(G):
        nc = pipe.pools[slot][lane].buffer("node_counter").cpu().numpy()
        a, n = int(nc[9 + H - 1]), int(nc[9 + H] - nc[9 + H - 1])
        spans.append((a, n))
        total += n
    ms_alone = sorted(pipe.regather_last(slot, repeats))
    out = {"rows_per_launch": total,
           "alone": {"avg_launch_us": ms_alone[len(ms_alone) // 2] * 1e3, "achieved": total * bytes_per_row / (ms_alone[len(ms_alone) // 2] * 1e-3) / 1e9}}
    out["alone"]["frac"] = out["alone"]["achieved"] / HBM_PEAK_GBPS
    if total > N:
        out["cold"] = None
        out["note"] = f"a launch gathers {total} rows, the graph has {N} vertices: no set of distinct rows of that size exists"
        return out
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    perm = torch.randperm(N, device=dev, generator=g)
    off = 0
    for lane, (a, n) in enumerate(spans):
        pl = pipe.pools[slot][lane]
        ids = perm[off:off + n].to(torch.int32)
        off += n
        pl.buffer("sampled_ids")[a:a + n] = ids
        if pl._lib.legion_pool_buffer(pl.handle, 13):                     # carried cache slots (column slots in use)
            pl.buffer("node_slot")[a:a + n] = node_map[ids.long()] if node_map.numel() > 0 else -3
    del perm
    torch.cuda.synchronize()
    ms_cold = sorted(pipe.regather_last(slot, repeats))
    med = ms_cold[len(ms_cold) // 2]
    out["cold"] = {"avg_launch_us": med * 1e3, "achieved": total * bytes_per_row / (med * 1e-3) / 1e9,
                   "frac": total * bytes_per_row / (med * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                   "frac_of_achievable": total * bytes_per_row / (med * 1e-3) / 1e9 / HBM_ACHIEVABLE_GBPS,
                   "achievable_peak": HBM_ACHIEVABLE_GBPS,
                   "min_launch_us": ms_cold[0] * 1e3, "max_launch_us": ms_cold[-1] * 1e3, "launches": len(ms_cold)}
    out["note"] = ("`alone`: the group's last-hop gather launched again with nothing beside it (inside a group the next group's head shares "
                   "the machine); `cold`: the same launch after every lane's last-hop ids were replaced by ids that do not repeat anywhere in "
                   "the launch (uniform over the vertices, cache slots rewritten to match): no row is read twice, nothing the 256 MiB "
                   "Infinity Cache could serve; medians of %d launches by HIP events" % repeats)
    return out


def committed_profiles(c, rows_last, n_last):
    """What the committed profiles of this configuration say about the same kernel: the PMC traffic figure rescaled by this run's rows
    (profiles/rNN/pmc_gather_kernel.json) and its average duration in the rocprofv3 kernel trace of the default command."""
    import glob
    args, G, B, D = c.args, c.G, c.B, c.D
    traffic, traffic_src = None, None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_gather_kernel.json")), reverse=True):
        try:
            pmc = json.load(open(f))
        except (OSError, ValueError):
            continue
        if pmc.get("batches_per_launch_group") == G and f"batch {B}," in pmc.get("config", "") and f"N x {D}]" in pmc.get("config", "") \
                and f"RMAT-{args.scale} " in pmc.get("config", "") and ("scrambled" in pmc.get("config", "")) == bool(args.scramble):
            traffic = pmc["traffic_bytes_per_launch"] / pmc["rows_per_launch"] * (rows_last / max(n_last, 1))
            traffic_src = os.path.relpath(f, ROOT)
            break
    rocprof_us, rocprof_src = None, None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "bench_default_kernel_trace_by_grid.csv")), reverse=True):
        try:
            for ln in open(f).read().splitlines()[1:]:
                cols = ln.rsplit(",", 10)       # kernel (its template arguments contain commas), then ten numeric columns
                if len(cols) == 11 and "gather_kernel" in cols[0] and cols[0].rstrip().endswith("true>") and int(cols[2]) == G:
                    rocprof_us, rocprof_src = float(cols[7]), os.path.relpath(f, ROOT)
                    break
        except (OSError, ValueError, KeyError):
            continue
        if rocprof_us is not None:
            break
    return traffic, traffic_src, rocprof_us, rocprof_src


def per_rank_info(c, timed, counted, achieved, t_last, n_last, pipe, bulk):
    """N > 1: what every rank saw, so that one SCALE invocation is its own evidence."""
    args, world, rank, D = c.args, c.world, c.rank, c.D
    lk0, lk1 = timed.lk0, timed.lk1
    window_s = timed.t_lk1 - timed.t_lk0
    mine_info = {"rank": rank, "pci_bus_id": lk1["pci_bus_id"], "gpu_metrics_revision": lk1["gpu_metrics_revision"],
                 "edges_per_sec": float(counted.edges.sum()) / max(timed.own_region, 1e-9),
                 "gather_roofline_frac": achieved / HBM_PEAK_GBPS, "gather_avg_launch_us": t_last / max(n_last, 1) * 1e6}
    if lk0["supported"] and lk1["supported"]:
        xr = lk1["xgmi_read_bytes"] - lk0["xgmi_read_bytes"]
        mine_info.update({"xgmi_read_bytes": xr, "xgmi_write_bytes": lk1["xgmi_write_bytes"] - lk0["xgmi_write_bytes"],
                          "xgmi_read_GBps": xr / max(window_s, 1e-9) / 1e9,
                          "xgmi_read_bytes_link": [b - a for a, b in zip(lk0["xgmi_read_bytes_link"], lk1["xgmi_read_bytes_link"])],
                          "pcie_bytes": lk1["pcie_bytes"] - lk0["pcie_bytes"], "window_s": window_s,
                          "window": f"{timed.repeats} timed regions incl. their barriers"})
    if counted.source_rows is not None:
        # rows of ONE timed region by where the gather read them: computed by the kernel in the untimed counting pass
        stripe_rows, replica_rows_read, peer_rows = counted.source_rows
        mine_info.update({"rows_from_own_stripe": stripe_rows - peer_rows, "rows_from_peer_stripes": peer_rows,
                          "rows_from_local_replica": replica_rows_read, "rows_gathered": int(counted.rows.sum()),
                          "peer_bytes_per_region_computed": peer_rows * D * 4,
                          "peer_read_GBps_computed": peer_rows * D * 4 / max(timed.own_region, 1e-9) / 1e9})
        if "xgmi_read_bytes" in mine_info:
            mine_info["xgmi_read_bytes_per_region_measured"] = mine_info["xgmi_read_bytes"] / timed.repeats
    if bulk:
        pushed = pipe.rows_listed                              # rows the other members pushed into this GPU per counted region
        mine_info["bulk"] = {"rows_pushed_into_me_per_region": pushed, "bytes_pushed_into_me_per_region": pushed * D * 4,
                             "phase_a_s_per_group": pipe.t_a / max(pipe.groups, 1), "phase_b_s_per_group": pipe.t_b / max(pipe.groups, 1),
                             "barriers_s_per_group": pipe.t_bar / max(pipe.groups, 1), "groups_clocked": pipe.groups,
                             "push_GBps_out_of_me": (pushed * D * 4 / max(args.steps, 1)) / max(pipe.t_b / max(pipe.groups, 1), 1e-9) / 1e9,
                             "note": "phase A = own sampler + per-owner lists + gather of local rows; phase B = this GPU as an owner pushing "
                                     "the rows the others listed (about as many as were pushed into it); wall clock incl. stream synchronise"}
    per_rank = [None] * world
    dist.all_gather_object(per_rank, mine_info)
    return per_rank


# the values of run_leg that leg_report reads (bench.py packs exactly these into the namespace it hands over)
REPORT_NAMES = frozenset(['B', 'D', 'G', 'H', 'achieved', 'args', 'bulk', 'bytes_per_row', 'c', 'cache', 'collective', 'counted', 'counters', 'd', 'edges', 'elapsed_max', 'elapsed_profiled', 'fanout', 'headline', 'hy_cpu', 'hy_gpu', 'last_op', 'lds_buckets', 'n_last', 'n_timed', 'payload_gbps', 'pcie_tx', 'per_rank', 'replica_memory', 'rocprof_src', 'rocprof_us', 'rows', 'rows_last', 'samp_bytes', 'setup_s', 'shape_leg', 'state_bytes', 'stripe', 't_all_gathers', 't_last', 't_sampling', 'timed', 'topo_tx', 'tot_edges', 'traffic_committed', 'traffic_committed_src', 'train_step', 'weave', 'world', 'xgmi_tx'])


def leg_report(v):
    """Rank 0's JSON report of a leg from the values run_leg measured (`v`: a namespace of them): the headline form (metric, value,
    config, roofline, sampling_only ...) or the extra legs' short form."""
    (B, D, G, H, achieved, args, bulk, bytes_per_row, c, cache, collective, counted, counters, d, edges, elapsed_max, elapsed_profiled, fanout,
     headline, hy_cpu, hy_gpu, last_op, lds_buckets, n_last, n_timed, payload_gbps, pcie_tx, per_rank, replica_memory, rocprof_src, rocprof_us,
     rows, rows_last, samp_bytes, setup_s, shape_leg, state_bytes, stripe, t_all_gathers, t_last, t_sampling, timed, topo_tx, tot_edges,
     traffic_committed, traffic_committed_src, train_step, weave, world, xgmi_tx) = (
        v.B, v.D, v.G, v.H, v.achieved, v.args, v.bulk, v.bytes_per_row, v.c, v.cache, v.collective, v.counted, v.counters, v.d, v.edges,
        v.elapsed_max, v.elapsed_profiled, v.fanout, v.headline, v.hy_cpu, v.hy_gpu, v.last_op, v.lds_buckets, v.n_last, v.n_timed, v.payload_gbps,
        v.pcie_tx, v.per_rank, v.replica_memory, v.rocprof_src, v.rocprof_us, v.rows, v.rows_last, v.samp_bytes, v.setup_s, v.shape_leg,
        v.state_bytes, v.stripe, v.t_all_gathers, v.t_last, v.t_sampling, v.timed, v.topo_tx, v.tot_edges, v.traffic_committed,
        v.traffic_committed_src, v.train_step, v.weave, v.world, v.xgmi_tx)
    layout = (f"seed-sharded x{world}, replicated graph+features, one clique of {world}: caches striped over the ranks, peer reads "
              f"over xGMI (cache_agg_mode {int(np.log2(world))})" + (f", hot-row replica of {replica_memory} bytes per GPU" if replica_memory else "")
              if stripe else f"seed-sharded x{world}, replicated graph+features, cache_agg_mode 0")
    roof = {"bound": "hbm", "kernel": "lg::gather_kernel<..., LASTOP = true> (hop-%d gather, op %d: the instance launched for a batch's last op)" % (H, last_op),
            "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS,
            "traffic": None, "traffic_unit": "bytes per launch", "traffic_source": None,
            "traffic_committed": traffic_committed, "traffic_committed_source": traffic_committed_src,
            "traffic_is": "`traffic` = HBM bytes per launch from FETCH_SIZE + WRITE_SIZE collected by child processes of THIS run (N = 1, "
                          "the default command; null where that leg did not run); `traffic_committed` = the committed profile's figure for "
                          "this configuration rescaled by this run's rows",
            "rocprofv3_avg_launch_us": rocprof_us, "rocprofv3_source": rocprof_src,
            "algorithmic_bytes_per_launch": rows_last / max(n_last, 1) * bytes_per_row,
            "bytes_per_row": bytes_per_row, "rows_per_launch": rows_last / max(n_last, 1),
            "launches": n_last, "avg_launch_us": t_last / max(n_last, 1) * 1e6,
            "measured": "HIP events on the launch stream around each hop-%d gather launch over the same %d "
                        "steps, %d batches per launch, eager launches (wall clock of that pass: ms_per_step %.4f; "
                        "hipGraph replay, timed region: %.4f)"
                        % (H, args.steps, G, elapsed_profiled / args.steps * 1e3, elapsed_max / args.steps * 1e3)}
    if counted.unique is not None:
        u = counted.unique
        roof["unique_row_frac"] = u["unique_rows"] / max(u["rows"], 1)
        roof["row_reuse"] = {"rows": u["rows"], "unique_rows": u["unique_rows"], "rows_gathered_once": u["rows_gathered_once"],
                             "repeat_rows": u["repeat_rows"], "unique_rows_bytes": u["unique_rows"] * D * 4,
                             "infinity_cache_bytes": INFINITY_CACHE_BYTES,
                             "note": "the last hop's rows of ONE launch group (%d mini-batches): a vertex new to several batches of the "
                                     "group is gathered once per batch; FETCH_SIZE counts a repeat the memory-side cache serves as a "
                                     "fabric read all the same (MI355X_MICROARCH.md), so `frac` is a fabric-side figure and "
                                     "`cold.frac` the HBM-only one" % G}
    timed_j = {"steps": args.steps, "repeats": timed.repeats, "median_s": elapsed_max,
               "min_s": float(timed.region_s.min()), "max_s": float(timed.region_s.max()), "total_timed_s": float(timed.region_s.sum()),
               "note": "exactly K steps per region between barrier+synchronize brackets; region repeated over the "
                       "same batches until --min-seconds; per repeat the max over ranks; value uses the median"}
    if not headline:
        out = {"value": float(tot_edges.item()) / elapsed_max, "unit": "edges/s", "ms_per_step": elapsed_max / args.steps * 1e3,
               "parallelism": layout, "timed_region": timed_j, "roofline": roof,
               "feature_gather_GBps": payload_gbps, "feature_cache_rows": cache.node_capacity(d),
               "topology_cache_vertices": cache.edge_capacity(d), "hot_row_replica_rows": cache.replica_rows(d),
               "collective": collective, "per_rank": per_rank, "setup_seconds": setup_s,
               "xgmi_ingest_peak_GBps_per_gpu": 7 * 153.0 / 2,
               "note": "same workload, same seed batches, same K steps as the headline; the feature and topology caches striped over "
                       "one clique of all ranks (hotness rank t on GPU t % N), remote rows and adjacency read with direct peer loads "
                       "over xGMI; per_rank[].rows_from_* were counted by the gather itself in an untimed pass over the timed batches, "
                       "xgmi_* are deltas of the driver's cumulative gpu_metrics counters over the timed regions"}
        if bulk:
            out["peer_gather"] = "bulk"
            out["note"] = ("same striped clique as `striped`, but the rows of other members' stripes are listed per owner and PUSHED by the "
                           "owners (LegionTuning.peer_gather = bulk: whole rows as coalesced posted stores over xGMI instead of scattered "
                           "512-1024-byte load round trips); eager launches, two host barriers per launch group (per_rank[].bulk has the "
                           "phase clocks): compare its xGMI GB/s and ms_per_step with `striped`, minus the barrier time")
        if shape_leg:
            for k in ("collective", "per_rank", "xgmi_ingest_peak_GBps_per_gpu", "hot_row_replica_rows"):
                out.pop(k, None)
            out.update({"steps": args.steps, "warmup": c.n_warm // G, "batches_per_step": G,
                        "workload": f"{c.graph_name}, float32[N x {D}] features, batch {B}, fanout {fanout}, all tables resident in HBM",
                        "sampling_only_edges_per_sec": float(edges.sum()) / t_sampling,
                        "edges_per_batch": float(edges.mean()), "rows_per_batch": float(rows.sum(axis=1).mean()),
                        "note": "another batch shape of the headline's graph and feature table, same code path, same brackets (K steps per "
                                "region, median region), in the same process right after the headline leg; not `value`"})
    else:
        out = {
            "metric": "sampled_edges_per_sec",
            "value": float(tot_edges.item()) / elapsed_max,
            "unit": "edges/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed_max / args.steps * 1e3,
            "batches_per_step": G, "ms_per_batch": elapsed_max / n_timed * 1e3,
            "timed_region": timed_j,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int32+f32(copy)", "data": "synthetic",
            "config": {"workload": f"{c.graph_name}, "
                                   f"float32[N x {D}] features, batch {B}, fanout {fanout}, " +
                                   ("all tables resident in HBM" if args.placement == "hbm" else
                                    "full CSR + full feature table in mapped pinned host memory (read over PCIe on a miss), "
                                    "hotness-ranked feature/topology caches in HBM")
                                   + (", vertex labels scrambled" if args.scramble else ""),
                       "parallelism": layout,
                       "batches_per_launch_group": G, "groups_in_flight": args.slots,
                       "lane_arrays": {True: "one arena of shuffled 2 MB physical chunks (LegionTuning.arena_scatter_mb)", "plain": "one plain arena", False: "separate allocations"}[args.lane_arena], "epoch_batches": c.epoch_batches,
                       "streams": "weave: head of group k+1 on a second stream under the heavy kernels of group k" if weave else "one",
                       "epochs_wrap": bool(c.wrap), "hipgraph": not args.no_graph,
                       "cache_memory_bytes": args.cache_memory,
                       "feature_cache_rows": cache.node_capacity(d), "topology_cache_vertices": cache.edge_capacity(d),
                       "presc_batches": train_step, "presc_topology_transactions": topo_tx,
                       "presc_pcie_transactions_gpu_metrics": pcie_tx, "presc_xgmi_transactions_gpu_metrics": xgmi_tx,
                       "link_counters": args.link_counters,
                       "cost_model_counters": list(counters),
                       "hot_row_replica_rows": cache.replica_rows(d)},
            "feature_gather_GBps": payload_gbps * 1.0,
            "feature_gather_GBps_note": "payload bytes read (rows*D*4) / HIP-event time of all gather launches, rank 0",
            "sampling_only": {"edges_per_sec": float(edges.sum()) / t_sampling, "algorithmic_GBps": samp_bytes / t_sampling / 1e9,
                              "frac_of_hbm_peak": samp_bytes / t_sampling / 1e9 / HBM_PEAK_GBPS,
                              "note": "rank 0; time = timed region minus the HIP-event time of all gather launches (with the weave "
                                      "arrangement the head of the next group runs hidden under this group's heavy kernels, so this is the "
                                      "sampler time that is NOT hidden); the sampling kernel is bound by the part's rate of random 128-byte requests, "
                                      "the de-duplication and compaction by their dependent chains: DESIGN.md section 4.2"},
            "edges_per_step": float(edges.sum()) / args.steps, "rows_per_step": float(rows.sum()) / args.steps,
            "edges_per_batch": float(edges.mean()), "rows_per_batch": float(rows.sum(axis=1).mean()),
            "seed_feature_cache_hits_step0": counted.hits,
            "roofline": roof,
            "setup_seconds": setup_s,
            "first_touch_state": {"form": "none per vertex: a hop's claims are de-duplicated bucket by bucket in LDS", "bytes_per_lane": state_bytes,
                                  "lanes": G * args.slots, "lds_buckets_per_lane": lds_buckets},
            "feature_cache_hit_rate": counted.feat_hit_rows / max(counted.feat_hit_rows + counted.feat_miss_rows, 1),
            "feature_cache_hit_rate_over": "every timed batch" if args.placement == "pinned" else "the first timed step",
        }
        if args.hybrid and counted.tiers is not None:
            t_cpu, t_gpu, t_miss = counted.tiers
            pcie_gbps = t_cpu / max(t_cpu + t_gpu + t_miss, 1) * float(rows.sum() * D * 4) / t_all_gathers / 1e9 if t_all_gathers > 0 else 0.0
            out["hybrid_tier"] = {"cpu_cache_rows": hy_cpu, "gpu_cache_rows": hy_gpu,
                                  "rows_from_cpu_cache": t_cpu, "rows_from_gpu_cache": t_gpu, "rows_from_table": t_miss,
                                  "rows_counted_over": "the first timed step",
                                  "cpu_cache_GBps_over_pcie": pcie_gbps, "pcie_peak_GBps": 64.0,
                                  "note": "UnifiedCache::HybridInit (SS/cache/cache.cu:614-670) instead of the cost model: the hottest "
                                          "gpu_cache_rows of this GPU's own order in HBM, the next cpu_cache_rows in mapped pinned host memory "
                                          "(read in place over PCIe inside the same gather launches), misses from the full table in HBM; "
                                          "cpu_cache_GBps = CPU-cache rows x D x 4 / HIP-event time of all gather launches"}
        if collective is not None:
            out["collective"] = collective
            out["per_rank"] = per_rank
        if args.placement == "pinned":
            miss_frac = counted.feat_miss_rows / max(counted.feat_hit_rows + counted.feat_miss_rows, 1)
            miss_gbps = float(rows.sum() * D * 4) * miss_frac / t_all_gathers / 1e9 if t_all_gathers > 0 else 0.0
            out["miss_path"] = {"feature_rows_missed_frac": miss_frac, "pcie_feature_GBps": miss_gbps,
                                "pcie_peak_GBps": 64.0, "frac_of_pcie_peak": miss_gbps / 64.0,
                                "note": "missed rows x D x 4 bytes / HIP-event time of all gather launches (hits are served from "
                                        "HBM inside the same launches); PCIe Gen5 x16 = 64 GB/s per direction; topology misses "
                                        "(4-byte column reads) cross the same link during the sampler kernels"}
    return out
