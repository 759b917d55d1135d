#!/usr/bin/env python3
"""bench.py -- sampled edges/s + feature-gather GB/s of the hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is ONE LAUNCH GROUP through the whole hot path on one GPU: `batches_per_step` (= --group,
default 512 at B = 1024) independent mini-batches, each seed batch -> 2-hop sampling [25,10] -> per-hop
feature-cache lookup + gather -> end-of-batch clean-up (the op order of the reference's
GPURunner::RunOnce, SS/engine/server.cu:302-332), served by one hipGraph replay, inputs resident in HBM.
The timed region is exactly K steps between barrier + synchronize brackets; because K steps of ~1 ms are
far too short to time (launch latency, clock ramp), the same region is repeated until >= --min-seconds of
GPU work have been timed and the MEDIAN region (max over ranks per repeat) gives `value`.
Workload: synthetic RMAT-26 (N = 2^26, E = 2^30), float32[N x 128] counter-hash
features, B = 1024, seeds = a seeded permutation, GPU p of P takes seeds with id % P == p.
Mini-batches are served in groups: every kernel launch covers --group (default 524288 / B, at most 512) independent
batches (grid.y = lanes) and a group's op list is one hipGraph replay (legion_amd/csrc/pipeline.hip).

One process per GPU.  The path shards by seeds with no per-batch exchange; the only collective is
the one-time all-reduce (RCCL) of the PreSC hotness counters that sizes the caches.  Weak scaling:
every rank runs K batches of B seeds; value = total sampled edges of all ranks / max-over-ranks time.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel, the hop-2 gather:
achieved = rows * (8*D + 8) bytes / HIP-event time around that launch, summed over the timed steps.
`cpu_baseline` is the oracle's C restatement (oracle/, test infrastructure) timed on the host cores
on a bounded sample of the same batches -- a reported baseline, not a target.
"""
import argparse
import ctypes
import json
import os
import signal
import subprocess
import sys
import threading
import time
import types

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md: 8 TB/s spec


class OneLine:
    """The ONE JSON line of rank 0, written exactly once.  At N > 1 the extra legs (striped caches over peer pointers) run after
    the headline leg and have never met more than one physical GPU before the driver's SCALE run: whatever ends this process
    while they run -- a Python error (caught by the caller), a native exit() of the library (libc atexit hook), SIGTERM from
    the launcher after another rank died (wake-up fd + watcher thread: works while the main thread is blocked inside a HIP or
    RCCL call), or nothing at all for `deadline_s` seconds (a hang) -- the headline leg's line, already measured, still goes
    out, with a note on what happened."""

    def __init__(self, fd):
        self.fd, self.line, self.done, self.lock = fd, None, False, threading.Lock()
        self._hook = None

    def emit(self, obj):
        with self.lock:
            if self.done:
                return
            self.done = True
        text = None
        for _ in range(5):          # (the fallback may serialise the object while the main thread adds a leg to it)
            try:
                text = json.dumps(obj)
                break
            except RuntimeError:
                time.sleep(0.01)
        if text is None:
            text = json.dumps({k: v for k, v in list(obj.items()) if k not in ("striped", "striped_replica", "striped_bulk")})
        os.write(self.fd, (text + "\n").encode())

    def arm(self, headline_obj, deadline_s=900):
        """From here on a dying -- or, after `deadline_s`, a hung -- process still prints `headline_obj` -- the object itself, not a
        copy: whatever extra legs have been added to it by then go out with it."""
        self.line = headline_obj

        def fallback(why):
            if self.line is not None and not self.done:
                self.line["extra_legs_error"] = why
                self.emit(self.line)

        # (glibc exports __cxa_atexit; plain atexit lives in its static part)
        self._hook = ctypes.CFUNCTYPE(None, ctypes.c_void_p)(lambda _: fallback("the process exited inside an extra leg (native exit)"))
        getattr(ctypes.CDLL(None), "__cxa_atexit")(self._hook, None, None)
        rfd, wfd = os.pipe()
        os.set_blocking(wfd, False)
        signal.set_wakeup_fd(wfd, warn_on_full_buffer=False)
        signal.signal(signal.SIGTERM, lambda *_: None)       # (the C-level handler writes the signal number to wfd)

        def watch():
            import select
            got, _, _ = select.select([rfd], [], [], deadline_s)
            fallback("SIGTERM while an extra leg was running (another rank failed?)" if got else
                     "the extra legs did not finish within %d s (hung peer load or collective?)" % deadline_s)
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
    ap.add_argument("--collective-deadline", type=int, default=120,
                    help="seconds the library's communicator may take to form before the run falls back to torch.distributed")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed and run the collectives even at N = 1 (exercises the RCCL calls on a 1-GPU box)")
    ap.add_argument("--extra-legs-deadline", type=int, default=600,
                    help="N > 1: seconds the extra (striped) legs may take together before rank 0 prints the headline line alone and exits")
    ap.add_argument("--fail-extra-leg", type=str, default="", choices=["", "raise", "exit", "sigterm", "hang"],
                    help="testing: make rank 0 fail this way when the first extra leg starts (the headline line must still go out)")
    ap.add_argument("--force-device", type=int, default=-1,
                    help="put every rank on this GPU (testing the N > 1 code path on a 1-GPU box, with --backend gloo)")
    return ap.parse_args()


def main():
    args = parse_args()
    # the library logs the reference's lines ("Alpha: ...", "Feat capacity: ...") on stdout from every
    # rank; keep the real stdout for the one JSON line and send everything else to stderr
    sys.stdout.flush()
    one_line = OneLine(os.dup(1))
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if args.force_device >= 0:
        local_rank = args.force_device
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    if int(os.environ.get("LOCAL_RANK", "0")) == 0:
        import __graft_entry__
        __graft_entry__.ensure_built()          # fresh checkout: build the in-tree artefacts once (rank 0)
    if use_dist:
        dist.barrier()
    from legion_amd import engine, synth

    c = types.SimpleNamespace(args=args, world=world, rank=rank, local_rank=local_rank, dev=dev, use_dist=use_dist)
    c.fanout = [int(x) for x in args.fanout.split(",")]
    c.H = len(c.fanout)
    c.N = N = args.nodes if args.nodes > 0 else 1 << args.scale
    c.D = D = args.dim
    c.B = B = args.batch
    c.t_setup = time.time()

    # ---- workload, resident in HBM --------------------------------------------------------------
    if args.gather_rows > 0:
        os.environ["LEGION_GATHER_ROWS"] = str(args.gather_rows)
    engine.set_device_base(local_rank)
    if args.nodes > 0:
        indptr, col = synth.csr_device_large(N, args.edges if args.edges > 0 else N * args.edge_factor, 20231, dev)
    else:
        indptr, col = synth.rmat_csr_device(args.scale, args.edge_factor, 20231, dev, scramble=args.scramble)
    torch.cuda.empty_cache()
    c.E = int(col.numel())
    c.graph_name = (f"RMAT-{int(np.ceil(np.log2(N)))} edges folded to N={N} vertices, E={c.E} (synth.csr_device_large)" if args.nodes > 0
                    else f"RMAT-{args.scale} EF{args.edge_factor} (N={N}, E={c.E})")
    c.pinned = []
    if args.placement == "pinned":
        # generate on the device, park in mapped pinned host memory; the device copies of the CSR stay for the checks
        indptr_hbm, col_hbm = indptr, col
        c.pinned = [engine.PinnedArray.empty((N + 1,), np.int64), engine.PinnedArray.empty((int(col.numel()),), np.int32),
                    engine.PinnedArray.empty((N, D), np.float32)]
        indptr, col, features = (p.tensor(dev) for p in c.pinned)
        indptr.copy_(indptr_hbm)
        col.copy_(col_hbm)
        for r0 in range(0, N, 1 << 22):
            features[r0:r0 + (1 << 22)].copy_(synth.features_device_rows(r0, min(1 << 22, N - r0), D, 7, dev))
        torch.cuda.synchronize()
    else:
        features = synth.features_device(N, D, 7, dev)
    c.indptr, c.col, c.features = indptr, col, features
    # mini-batches per step (launch group): 524288 // B rounded down to a power of two, at most 512 (512 at B = 1024, 64 at
    # B = 8000) -- measured on one box: 128 / 256 / 512 / 1024 lanes 5.04 / 5.27 / 5.43-5.52 / 5.52-5.56 G edges/s at B = 1024,
    # 32 / 64 lanes 5.67 / 5.83 G at B = 8000 -- halved while the lanes of all groups in flight would take more than 0.7
    # of the HBM that the tables left free
    if args.group > 0:
        G = args.group
    else:
        G = 1
        while G * 2 <= 512 and G * 2 * B <= 524288:
            G *= 2
        num_ids, per = B, B
        for f in c.fanout:
            per *= f
            num_ids += per
        lane_bytes = num_ids * 56 + per * 28 + (num_ids // (8 if c.H <= 2 else 16)) * D * 4   # ids / edges / headers, slot arrays, feature rows (1.2 x the unique nodes: ~1/8 of num_ids at two hops, less beyond)
        free_t = torch.tensor([torch.cuda.mem_get_info(dev)[0]], dtype=torch.int64, device=dev)
        if use_dist:
            dist.all_reduce(free_t, op=dist.ReduceOp.MIN)                      # every rank takes the same group size
        while G > 1 and G * args.slots * lane_bytes > int(free_t.item()) * 7 // 10:
            G //= 2
    c.G = G
    c.n_warm, c.n_timed = args.warmup * G, args.steps * G                      # in mini-batches
    need = (c.n_warm + c.n_timed + 2) * B * world + B
    need = max(need, (args.presc_steps + 2) * B * world)
    all_seeds = synth.seed_ids(N, min(max(need * 2, N // 10), N), 11)
    c.mine = np.ascontiguousarray(all_seeds[all_seeds % world == rank])      # storage_management.cu:178
    # an epoch = the whole groups this rank's seed set holds; a run longer than that starts another epoch over the same
    # seeds (the reference's schedule wraps the same way: GetLocalBatchId, ipc_service.cu:213-228)
    c.epoch_batches = ((c.mine.size - 1) // B) // G * G
    assert c.epoch_batches >= G, f"the seed set of this rank ({c.mine.size} ids) holds less than one launch group of {G} x {B}"
    c.wrap = c.epoch_batches if c.n_warm + c.n_timed > c.epoch_batches else None

    # ---- the headline leg, then (N > 1) the same workload with the caches striped over one clique of N ------------
    stripe = args.stripe and world > 1
    head = run_leg(c, engine, synth, stripe, args.replica_memory if stripe else 0, headline=True)
    out = head["json"] if rank == 0 else None
    if world > 1 and not stripe and not args.no_striped_leg:
        if rank == 0:
            one_line.arm(out, args.extra_legs_deadline)
        try:
            out_striped = run_leg(c, engine, synth, True, 0, headline=False)
            if rank == 0:
                out["striped"] = out_striped["json"]
            if args.striped_replica_memory > 0:
                out_rep = run_leg(c, engine, synth, True, args.striped_replica_memory, headline=False)
                if rank == 0:
                    out["striped_replica"] = out_rep["json"]
            if not args.no_bulk_leg:
                out_bulk = run_leg(c, engine, synth, True, 0, headline=False, bulk=True)
                if rank == 0:
                    out["striped_bulk"] = out_bulk["json"]
        except Exception as e:      # the headline stands; say what the extra leg did
            import traceback
            traceback.print_exc()
            if rank == 0:
                out["extra_legs_error"] = f"{type(e).__name__}: {e}"[:600]
                one_line.emit(out)
            # not `raise`: the armed libc exit hook is a Python callable, and an interpreter that finalises first would have
            # libc call into it afterwards (segfault / changed exit status).  Leave without finalising, as the success path does.
            sys.stdout.flush()
            sys.stderr.flush()
            os._exit(1)

    if rank == 0:
        try:
            if world == 1 and not args.no_boundary and args.placement == "hbm":
                out.update(boundary_leg(args, c.fanout))
            if args.cpu_seconds > 0 and world == 1:      # reported at N = 1 only
                out["cpu_baseline"] = cpu_baseline(indptr, col, c.mine, N, B, c.fanout, c.n_warm, args.cpu_seconds,
                                                   features if args.placement == "hbm" else None)
            if world == 1 and not args.no_boundary and not args.no_traffic_leg and args.placement == "hbm":
                measured_traffic(args, out["roofline"], c.G)
        except Exception as e:      # (an armed exit hook must not outlive the interpreter: see above)
            import traceback
            traceback.print_exc()
            out["post_legs_error"] = f"{type(e).__name__}: {e}"[:600]
            one_line.emit(out)
            sys.stderr.flush()
            os._exit(1)
        one_line.emit(out)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if one_line.line is not None:     # armed: the libc exit hook is a Python callable and must not outlive the interpreter
        sys.stderr.flush()
        os._exit(0)


def run_leg(c, engine, synth, stripe, replica_memory, headline, bulk=False):
    """One cache layout over the resident workload: objects -> PreSC -> hotness all-reduce -> cost model -> fills -> pipeline ->
    counting pass -> warm-up -> timed regions -> eager pass with HIP events around the gathers.  Returns {"json": rank 0's
    report of the leg}.  The headline leg also verifies; the extra legs are shorter."""
    args, world, rank, dev, use_dist = c.args, c.world, c.rank, c.dev, c.use_dist
    if not headline and args.fail_extra_leg and rank == 0:        # (tests of OneLine)
        if args.fail_extra_leg == "raise":
            raise RuntimeError("requested by --fail-extra-leg")
        if args.fail_extra_leg == "exit":
            ctypes.CDLL(None).exit(3)
        if args.fail_extra_leg == "sigterm":
            os.kill(os.getpid(), signal.SIGTERM)
        while True:
            ctypes.CDLL(None).sleep(5)      # (a main thread that never comes back from a C call)
    fanout, H, N, D, B, G = c.fanout, c.H, c.N, c.D, c.B, c.G
    n_warm, n_timed, wrap, mine = c.n_warm, c.n_timed, c.wrap, c.mine
    t_leg = time.time()
    P = world if stripe else 1              # logical GPUs the objects know about
    d = rank if stripe else 0               # the one this process owns
    engine.set_device_base(max(c.local_rank - d, 0))   # logical GPU d of this process = physical GPU LOCAL_RANK
    engine.set_local_device(d if stripe else -1)
    red_dev = dev if args.backend == "nccl" else "cpu"

    graph = engine.GraphStorage(P, c.indptr, c.col)
    feature = engine.FeatureStorage(P, c.features)
    feature.set_ids(d, engine.TRAINMODE, mine, None)
    train_step = min((mine.size - 1) // B, args.presc_steps)
    if use_dist:                            # train_step = min over partitions (ipc_service.cu:73-82)
        ts = torch.tensor([train_step], device=dev)
        dist.all_reduce(ts, op=dist.ReduceOp.MIN)
        train_step = int(ts.item())
    cache = engine.UnifiedCache(args.cache_memory, D, train_step, P, N)
    cache.init_controller(d)
    pool = engine.MemoryPool(d, N, B, fanout, D, pipeline_depth=1)

    # ---- PreSC epoch (bounded) -> hotness -> all-reduce over ranks -> order -> cost model -> fills --
    if args.measured_counters and args.link_counters == "v2":
        args.link_counters = "computed"
    torch.cuda.synchronize()
    time.sleep(0.02)
    lc0 = engine.link_counters(d)
    for it in range(train_step):
        engine.enqueue_batch(None, graph, feature, cache, pool, B, it, d, engine.TRAINMODE, True, fanout)
    torch.cuda.synchronize()
    time.sleep(0.02)
    lc1 = engine.link_counters(d)
    pcie_tx = (lc1[0] - lc0[0]) // 64 if (lc0 is not None and lc1 is not None) else None
    xgmi_tx = (lc1[1] - lc0[1]) // 64 if (lc0 is not None and lc1 is not None) else None
    collective = None
    if use_dist:    # the only collective of the path: RCCL all-reduce of the uint64 hotness counters
        ones = torch.ones(1, dtype=torch.int64, device=red_dev)
        dist.all_reduce(ones)                                   # the world size as torch.distributed sees it
        torch.cuda.synchronize()
        # The PRODUCT issues the collective (legion_amd/csrc/collective.hip: ncclAllReduce(ncclUint64, ncclSum) over its own
        # communicator); torch.distributed only carries rank 0's 128-byte unique id to the other ranks.  A join or a call that
        # fails or does not return within --collective-deadline seconds falls back to dist.all_reduce on the same arrays and
        # the line says so -- a SCALE run must not be lost to the first meeting of this code with a second physical GPU.
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
                world_seen, ar_ms_lib = cache.allreduce_hotness(d)
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
        collective = {"backend": "rccl" if "liblegion" in issued_by else dist.get_backend(), "issued_by": issued_by,
                      "world_size_seen_by_all_reduce": int(world_seen),
                      "hotness_all_reduce_ms": float(ar_t.item()), "hotness_all_reduce_bytes": 2 * N * 8,
                      "hotness_all_reduce_GBps_algorithmic": 2 * N * 8 / max(float(ar_t.item()), 1e-6) / 1e6,
                      "note": "two uint64[N] arrays (node and edge access counts), all-reduced in place once before CandidateSelection; "
                              "time = max over ranks, wall clock around both calls incl. synchronize"}
    max_ids = cache.max_id_num(d)
    topo_tx = cache.topo_transactions(d)
    if use_dist:
        tt = torch.tensor([topo_tx], dtype=torch.int64, device=red_dev)
        dist.all_reduce(tt)
        topo_tx = int(tt.item())
    if args.link_counters == "smi" and pcie_tx is not None:
        tt = torch.tensor([pcie_tx, xgmi_tx], dtype=torch.int64, device=red_dev)
        if use_dist:
            dist.all_reduce(tt)
        counters = (int(tt[0].item()), int(tt[1].item()))
    elif args.link_counters in ("computed", "smi"):
        counters = (topo_tx, 0)
    else:
        counters = (0, 0)
    if stripe:
        mids = [None] * world
        dist.all_gather_object(mids, max_ids)
        cache.set_peer_max_ids(mids)
        cache.candidate_selection(int(np.log2(world)), graph, world_reduced=True)
        cache.cost_model(feature, graph, counters, train_step)

        def all_gather_bytes(b):
            out = [None] * world
            dist.all_gather_object(out, b)
            return out

        if replica_memory > 0:
            cache.set_replica_memory(replica_memory)
        cache.fill_up_distributed(feature, graph, d, world, mids, all_gather_bytes)
        dist.barrier()
    else:
        cache.candidate_selection(0, graph, world_reduced=use_dist)
        cache.cost_model(feature, graph, counters, train_step)
        if args.capacity:
            cache.set_capacity(*[int(x) for x in args.capacity.split(",")])
        if not args.no_cache:
            cache.fill_up(feature, graph)
    feature_rows = int(max_ids * 1.2)                                        # server.cu:277
    pool.close()
    weave = not (args.no_weave or args.overlap)
    if bulk:
        # peer_gather = bulk (pipeline.hip): the rows of other members' stripes are pushed by their OWNERS; a group runs as
        # phase A (own sampler + lists + local gather) -> barrier -> phase B (push for the others) -> barrier, eager launches
        weave = False
        pipe = BulkPipe(engine.Pipeline(graph, feature, cache, d, B, fanout, G, feature_rows, False, args.slots, arena="shared"), use_dist)
        hs = [None] * world
        dist.all_gather_object(hs, pipe.p.bulk_export())
        for r, h in enumerate(hs):
            if r != rank:
                pipe.p.bulk_import(h)
        dist.barrier()
    else:
        pipe = engine.Pipeline(graph, feature, cache, d, B, fanout, G, feature_rows, not args.no_graph, args.slots,
                               args.overlap, False, weave, arena=args.lane_arena)
    torch.cuda.synchronize()
    setup_s = time.time() - (c.t_setup if headline else t_leg)

    # ---- untimed counting pass over exactly the timed batches (deterministic) --------------------
    first = n_warm
    edges = np.zeros(n_timed, dtype=np.int64)
    rows = np.zeros((n_timed, H + 1), dtype=np.int64)
    hop_edges = np.zeros((n_timed, H), dtype=np.int64)
    hop_slots = np.zeros((n_timed, H), dtype=np.int64)
    hits = 0
    have_map = cache.node_capacity(d) > 0 and not args.no_cache
    node_map = cache.array("node_map", d) if have_map else torch.empty(0, dtype=torch.int32, device=dev)
    feat_hit_rows = feat_miss_rows = 0           # over every timed batch (all hops)
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
        edges[k] = ec[9 + H]
        rows[k, 0] = nc[9]
        if node_map.numel() > 0 and (args.placement == "pinned" or k < G):
            hm = node_map[pl.buffer("sampled_ids")[:int(nc[9 + H])].long()] >= 0
            feat_hit_rows += int(hm.sum())
            feat_miss_rows += int(hm.numel() - int(hm.sum()))
        for h in range(H):
            rows[k, h + 1] = nc[9 + h + 1] - nc[9 + h]
            hop_edges[k, h] = ec[9 + h + 1] - ec[9 + h]
            hop_slots[k, h] = (nc[9] if h == 0 else ec[9 + h] - ec[9 + h - 1]) * fanout[h]
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
            hits = int((pl.buffer("cache_search_buffer")[:int(nc[1])] >= 0).sum())
    if bulk:
        pipe.count_rows = False
        pipe.reset_clocks()
    source_rows = None
    if stripe:                                   # where this rank's gathers read the timed batches' hit rows from
        source_rows = cache.gather_stats3(d)
        cache.gather_stats_enable(False)         # counting costs an atomic per hit row: off before anything is timed

    # ---- warm-up, then the timed region: exactly K steps (K hipGraph replays of G batches each) between
    #      barrier + synchronize brackets.  The region is repeated (same batches: an epoch over the same
    #      seeds, replays are deterministic) until --min-seconds have been timed; every rank runs the same
    #      number of repeats, per repeat the MAX over ranks counts, and the median repeat is reported. -------
    pipe.run_range(0, n_warm, wrap=wrap)
    pipe.wait()

    def timed_region(p=None):
        p = p or pipe
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        last = p.run_range(first, n_timed, wrap=wrap)
        p.wait()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        return time.perf_counter() - t0, last

    min_seconds = args.min_seconds if headline else 0.5 * args.min_seconds
    time.sleep(0.005)
    lk0, t_lk0 = engine.link_counters_ex(d), time.perf_counter()
    el0, last_group = timed_region()
    reps_t = torch.tensor([max(1, min(args.max_repeats, int(np.ceil(min_seconds / max(el0, 1e-6)))))],
                          dtype=torch.int64, device=dev)
    if use_dist:
        dist.all_reduce(reps_t, op=dist.ReduceOp.MAX)
    repeats = int(reps_t.item())
    region_s = [el0]
    for _ in range(repeats - 1):
        el, last_group = timed_region()
        region_s.append(el)
    time.sleep(0.005)
    lk1, t_lk1 = engine.link_counters_ex(d), time.perf_counter()
    own_region = float(np.median(np.asarray(region_s)))                     # this rank's own clock (brackets include the barriers)
    region_t = torch.tensor(region_s, dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(region_t, op=dist.ReduceOp.MAX)              # per repeat: the slowest rank
    region_s = region_t.cpu().numpy()
    elapsed_max = float(np.median(region_s))
    elapsed = elapsed_max                                            # used for the rank-0 sampler/gather split below
    if headline and not args.no_verify and last_group is not None:
        # the last group of the timed region is still in its slot: its batches must be the ones the counting
        # pass saw (replays are deterministic), and its last lane passes the full-size property checks
        slot, k0, n_lanes = last_group
        for lane in range(n_lanes):
            pl = pipe.pools[slot][lane]
            nc = pl.buffer("node_counter").cpu().numpy()
            ec = pl.buffer("edge_counter").cpu().numpy()
            k = n_timed - n_lanes + lane          # the last group submitted holds the last n_lanes batches of the region
            assert ec[9 + H] == edges[k] and nc[9 + H] - nc[9] == rows[k, 1:].sum(), f"replayed batch {k0 + lane} differs"
        n = int(nc[9 + H])
        ids = pl.buffer("sampled_ids")[:n]
        assert synth.feature_check_device(pl.buffer("float_features")[:n].contiguous(), ids.contiguous(), D, 7) == 0
        assert int(torch.unique(ids).numel()) == n

    # ---- the same K batches once more with HIP events around every gather launch (recorded on the
    #      lane's own stream).  Eager launches: HIP cannot time events recorded by graph nodes. ------
    pipe.profile_begin()
    pipe.run_range(0, n_warm, wrap=wrap)
    pipe.wait()
    warm = pipe.profile_read()
    t1 = time.perf_counter()
    pipe.run_range(first, n_timed, wrap=wrap)
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
    state_bytes, lds_buckets = pipe.pools[0][0].state_bytes(), pipe.pools[0][0].lds_buckets()

    pipe.close()

    tot_edges = torch.tensor([float(edges.sum())], dtype=torch.float64, device=dev)
    gather_bytes_t = torch.tensor([float(rows.sum() * D * 4)], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(tot_edges)
        dist.all_reduce(gather_bytes_t)

    # ---- roofline of the dominant kernel: the last hop's gather (op 3H+1) -------------------------
    last_op = 3 * H + 1
    t_last, n_last = prof.get(last_op, (0.0, 0))
    t_last *= 1e-3
    rows_last = int(rows[:, H].sum())
    bytes_per_row = 8 * D + 8
    achieved = rows_last * bytes_per_row / t_last / 1e9 if t_last > 0 else 0.0
    t_all_gathers = sum(v[0] for v in prof.values()) * 1e-3
    payload_gbps = float(rows.sum() * D * 4) / t_all_gathers / 1e9 if t_all_gathers > 0 else 0.0
    # sampler side: bytes = sum_h [S_h*25/f_h + E_h*44 + U_h*8]; time = step time minus the gathers
    samp_bytes = sum(float(hop_slots[:, h].sum()) * 25.0 / fanout[h] + float(hop_edges[:, h].sum()) * 44.0 +
                     float(rows[:, h + 1].sum()) * 8.0 for h in range(H))
    t_sampling = max(elapsed - t_all_gathers, 1e-9)

    # ---- N > 1: what every rank saw, so that one SCALE invocation is its own evidence ------------------------------
    per_rank = None
    if use_dist:
        window_s = t_lk1 - t_lk0
        mine_info = {"rank": rank, "pci_bus_id": lk1["pci_bus_id"], "gpu_metrics_revision": lk1["gpu_metrics_revision"],
                     "edges_per_sec": float(edges.sum()) / max(own_region, 1e-9),
                     "gather_roofline_frac": achieved / HBM_PEAK_GBPS, "gather_avg_launch_us": t_last / max(n_last, 1) * 1e6}
        if lk0["supported"] and lk1["supported"]:
            xr = lk1["xgmi_read_bytes"] - lk0["xgmi_read_bytes"]
            mine_info.update({"xgmi_read_bytes": xr, "xgmi_write_bytes": lk1["xgmi_write_bytes"] - lk0["xgmi_write_bytes"],
                              "xgmi_read_GBps": xr / max(window_s, 1e-9) / 1e9,
                              "xgmi_read_bytes_link": [b - a for a, b in zip(lk0["xgmi_read_bytes_link"], lk1["xgmi_read_bytes_link"])],
                              "pcie_bytes": lk1["pcie_bytes"] - lk0["pcie_bytes"], "window_s": window_s,
                              "window": f"{repeats} timed regions incl. their barriers"})
        if source_rows is not None:
            # rows of ONE timed region by where the gather read them: computed by the kernel in the untimed counting pass
            stripe_rows, replica_rows_read, peer_rows = source_rows
            mine_info.update({"rows_from_own_stripe": stripe_rows - peer_rows, "rows_from_peer_stripes": peer_rows,
                              "rows_from_local_replica": replica_rows_read, "rows_gathered": int(rows.sum()),
                              "peer_bytes_per_region_computed": peer_rows * D * 4,
                              "peer_read_GBps_computed": peer_rows * D * 4 / max(own_region, 1e-9) / 1e9})
            if "xgmi_read_bytes" in mine_info:
                mine_info["xgmi_read_bytes_per_region_measured"] = mine_info["xgmi_read_bytes"] / repeats
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

    # HBM traffic of that kernel cannot be read live: it comes from the committed PMC summary
    # (profiles/rNN/pmc_gather_kernel.json: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of
    # this same command, gfx950 corrections applied), used only when it was taken on this configuration
    traffic, traffic_src = None, None
    import glob
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

    # the same kernel's average duration in the committed rocprofv3 kernel trace of this command (profiles/rNN/, tools/profile_round.sh)
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

    out = None
    if rank == 0:
        layout = (f"seed-sharded x{world}, replicated graph+features, one clique of {world}: caches striped over the ranks, peer reads "
                  f"over xGMI (cache_agg_mode {int(np.log2(world))})" + (f", hot-row replica of {replica_memory} bytes per GPU" if replica_memory else "")
                  if stripe else f"seed-sharded x{world}, replicated graph+features, cache_agg_mode 0")
        roof = {"bound": "hbm", "kernel": "lg::gather_kernel<..., LASTOP = true> (hop-%d gather, op %d: the instance launched for a batch's last op)" % (H, last_op),
                "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_unit": "bytes per launch",
                "traffic_source": traffic_src, "traffic_committed_source": traffic_src,
                "traffic_is": "`traffic` = the committed PMC figure of this kernel on this configuration rescaled by this run's rows (a claim "
                              "about the kernel; counters cannot be read while timing); `traffic_measured` (N = 1, when the traffic leg ran) "
                              "= the same two counters collected by child processes of THIS run",
                "rocprofv3_avg_launch_us": rocprof_us, "rocprofv3_source": rocprof_src,
                "algorithmic_bytes_per_launch": rows_last / max(n_last, 1) * bytes_per_row,
                "bytes_per_row": bytes_per_row, "rows_per_launch": rows_last / max(n_last, 1),
                "launches": n_last, "avg_launch_us": t_last / max(n_last, 1) * 1e6,
                "measured": "HIP events on the launch stream around each hop-%d gather launch over the same %d "
                            "steps, %d batches per launch, eager launches (wall clock of that pass: ms_per_step %.4f; "
                            "hipGraph replay, timed region: %.4f)"
                            % (H, args.steps, G, elapsed_profiled / args.steps * 1e3, elapsed_max / args.steps * 1e3)}
        timed = {"steps": args.steps, "repeats": repeats, "median_s": elapsed_max,
                 "min_s": float(region_s.min()), "max_s": float(region_s.max()), "total_timed_s": float(region_s.sum()),
                 "note": "exactly K steps per region between barrier+synchronize brackets; region repeated over the "
                         "same batches until --min-seconds; per repeat the max over ranks; value uses the median"}
        if not headline:
            out = {"value": float(tot_edges.item()) / elapsed_max, "unit": "edges/s", "ms_per_step": elapsed_max / args.steps * 1e3,
                   "parallelism": layout, "timed_region": timed, "roofline": roof,
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
        else:
            out = {
                "metric": "sampled_edges_per_sec",
                "value": float(tot_edges.item()) / elapsed_max,
                "unit": "edges/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": elapsed_max / args.steps * 1e3,
                "batches_per_step": G, "ms_per_batch": elapsed_max / n_timed * 1e3,
                "timed_region": timed,
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
                           "epochs_wrap": bool(wrap), "hipgraph": not args.no_graph,
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
                "seed_feature_cache_hits_step0": hits,
                "roofline": roof,
                "setup_seconds": setup_s,
                "first_touch_state": {"form": "none per vertex: a hop's claims are de-duplicated bucket by bucket in LDS", "bytes_per_lane": state_bytes,
                                      "lanes": G * args.slots, "lds_buckets_per_lane": lds_buckets},
                "feature_cache_hit_rate": feat_hit_rows / max(feat_hit_rows + feat_miss_rows, 1),
                "feature_cache_hit_rate_over": "every timed batch" if args.placement == "pinned" else "the first timed step",
            }
            if collective is not None:
                out["collective"] = collective
                out["per_rank"] = per_rank
            if args.placement == "pinned":
                miss_frac = feat_miss_rows / max(feat_hit_rows + feat_miss_rows, 1)
                miss_gbps = float(rows.sum() * D * 4) * miss_frac / t_all_gathers / 1e9 if t_all_gathers > 0 else 0.0
                out["miss_path"] = {"feature_rows_missed_frac": miss_frac, "pcie_feature_GBps": miss_gbps,
                                    "pcie_peak_GBps": 64.0, "frac_of_pcie_peak": miss_gbps / 64.0,
                                    "note": "missed rows x D x 4 bytes / HIP-event time of all gather launches (hits are served from "
                                            "HBM inside the same launches); PCIe Gen5 x16 = 64 GB/s per direction; topology misses "
                                            "(4-byte column reads) cross the same link during the sampler kernels"}
    # this leg's objects go before the next leg builds its own (other logical-GPU numbering, other cache layout)
    cache.close()
    feature.close()
    graph.close()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    return {"json": out}


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
    """roofline.traffic MEASURED in this run: two fresh child processes of this very command -- started as
    children, never an exec of this process -- under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `... WRITE_SIZE` (separate
    passes, counters beside the kernel trace only, the interpreter binary directly behind `--`: MI355X_MICROARCH.md, HBM), two
    timed steps each on the same workload and group size; folded as tools/pmc_summary.py folds the committed profile (both
    counters KiB; FETCH_SIZE doubled: gfx950 tallies the 128-byte requests of a 16-byte-per-lane stream at 64 B).  Adds
    traffic_measured / traffic_over_algorithmic / traffic_source to `roof`; the committed figure stays beside it as the
    cross-check.  A child that fails or hangs costs only these fields."""
    import csv
    import glob
    import shutil
    import tempfile
    prof = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if prof is None:
        roof["traffic_measured"], roof["traffic_measured_note"] = None, "rocprofv3 not found on this box"
        return
    t0 = time.time()
    shape = ["--scale", str(args.scale), "--edge-factor", str(args.edge_factor), "--dim", str(args.dim), "--batch", str(args.batch),
             "--fanout", args.fanout, "--group", str(G), "--cache-memory", str(args.cache_memory)]
    if args.nodes > 0:
        shape += ["--nodes", str(args.nodes), "--edges", str(args.edges)]
    if args.scramble:
        shape += ["--scramble"]
    child = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--presc-steps", "64", "--cpu-seconds", "0",
             "--no-verify", "--no-boundary", "--min-seconds", "0.01"] + shape
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
        roof["traffic_measured"], roof["traffic_measured_note"] = None, note or "incomplete"
        return
    read_b, write_b = 2 * got["FETCH_SIZE"][0] * 1024, got["WRITE_SIZE"][0] * 1024
    alg = rows_child * roof["bytes_per_row"]
    roof["traffic_measured"] = read_b + write_b
    roof["traffic_measured_read_bytes"], roof["traffic_measured_write_bytes"] = read_b, write_b
    roof["traffic_over_algorithmic"] = (read_b + write_b) / alg
    roof["traffic_source"] = "this run"
    roof["traffic_measured_note"] = ("bytes per launch of the last hop's gather over a full group, averaged over %d / %d launches of two child "
                                     "processes of this command under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (%d rows per launch there; "
                                     "FETCH_SIZE doubled per the guide's gfx950 correction, both KiB); %.0f s" %
                                     (got["FETCH_SIZE"][1], got["WRITE_SIZE"][1], rows_child, time.time() - t0))


def cpu_baseline(indptr, col, seeds, N, B, fanout, first_batch, target_s, features=None):
    """The oracle's sampler (same Legion semantics, same batches) on the host cores: `value` is sampling only;
    `with_gather` adds the oracle's feature gather over a host copy of the table when the host has the memory for it."""
    from oracle import ffi
    L = ffi.load()
    ip = indptr.cpu().numpy()
    cl = col.cpu().numpy()
    g = ffi.OracleGraph(1, ip, cl)
    cores = len(os.sched_getaffinity(0))
    fan = np.asarray(fanout, dtype=np.int32)
    sd = np.ascontiguousarray(seeds, dtype=np.int32)

    def timed(nb, threads=cores, feats=None):
        secs = ctypes.c_double(0)
        nodes = ctypes.c_int64(0)
        e = L.lgo_bench_batches(ctypes.byref(g.g), N, ffi._p(sd, ffi.P_I32), int(sd.size), B, ffi._p(fan, ffi.P_I32),
                                len(fanout), first_batch, nb, threads,
                                feats.ctypes.data_as(ffi.P_F32) if feats is not None else None,
                                int(feats.shape[1]) if feats is not None else 0, ctypes.byref(secs), ctypes.byref(nodes))
        return int(e), secs.value, int(nodes.value)

    dgl_note = "DGL unavailable on this box (import dgl failed): oracle/dgl_semantics.c restates its sampler's semantics (SURVEY.md A.9)"
    try:
        import dgl  # noqa: F401
        dgl_note = "dgl importable but not used here; oracle/dgl_semantics.c restates its sampler's semantics (SURVEY.md A.9)"
    except Exception:
        pass
    avail = (sd.size - 1) // B - first_batch

    def sized(run_nb, budget_s):
        """calibrate on one batch per thread, then size the sample for budget_s seconds (bounded by the seed set); the
        calibration pays the threads' start-up (each allocates its own scratch), so the sample is re-sized once from its own rate"""
        e0, s0 = run_nb(min(cores, avail))
        nb_ = int(max(cores, cores * budget_s / max(s0, 1e-3)))
        nb_ = max(1, min(nb_, avail))
        e_, s_ = run_nb(nb_)
        for _ in range(2):
            if s_ >= 0.85 * budget_s or nb_ >= avail:
                break
            nb_ = max(1, min(avail, int(nb_ * budget_s / max(s_, 1e-3))))
            e_, s_ = run_nb(nb_)
        return e_, s_, nb_

    e, s, nb = sized(lambda n: timed(n)[:2], target_s)
    e1, s1, _ = timed(4, 1)                   # the same sampler on one thread (SURVEY 8d asks for both)
    e1, s1, _ = timed(int(max(4, min(256, 2.0 * 4 / max(s1, 1e-3)))), 1)
    out = {"value": e / s, "unit": "edges/s", "cores": cores, "kind": "port",
           "single_thread_edges_per_sec": e1 / s1,
           "sample": f"{nb} batches of {B} seeds (same RMAT graph, same fan-out, Legion semantics = the parity oracle's sampler, "
                     f"sampling only, no gather), {s:.1f} s on {cores} threads",
           "dgl": dgl_note}

    # DGL NeighborSampler semantics (without replacement, de-duplicated frontier, per-hop blocks) on the same seed batches:
    # the baseline north_star names; its edge count is its own, so is its edges/s
    def timed_dgl(nb_, threads=cores):
        secs = ctypes.c_double(0)
        nodes = ctypes.c_int64(0)
        ed_ = L.lgo_dgl_bench_batches(ffi._p(ip, ffi.P_I64), ffi._p(cl, ffi.P_I32), ffi._p(sd, ffi.P_I32), int(sd.size), B,
                                      ffi._p(fan, ffi.P_I32), len(fanout), first_batch, nb_, threads, ctypes.byref(secs), ctypes.byref(nodes))
        return int(ed_), secs.value, int(nodes.value)

    try:
        ed, sdg, nbd = sized(lambda n: timed_dgl(n)[:2], target_s)
        ed1, sd1, _ = timed_dgl(int(max(4, min(256, 2.0 * 4 / max(timed_dgl(4, 1)[1], 1e-3)))), 1)
        out["dgl_semantics"] = {"value": ed / sdg, "unit": "edges/s", "cores": cores, "kind": "dgl-semantics port",
                                "single_thread_edges_per_sec": ed1 / sd1, "edges_per_batch": ed / nbd,
                                "sample": f"{nbd} batches of {B} seeds (same graph, same seed batches, fan-out {fanout} from the seeds outward, "
                                          f"uniform WITHOUT replacement over the de-duplicated frontier, per-hop blocks; sampling only), "
                                          f"{sdg:.1f} s on {cores} threads",
                                "note": "oracle/dgl_semantics.c: a restatement of dgl.dataloading.NeighborSampler's semantics, not DGL's "
                                        "code (not installable here); a throughput reference, never a parity oracle"}
    except Exception as ex:       # reported baseline only: never fail the bench over it
        out["dgl_semantics"] = {"error": repr(ex)[:200]}
    # the whole path on the CPU (sampling + the oracle's row gather) when a host copy of the table is affordable
    try:
        import psutil
        table_bytes = int(features.numel()) * 4 if features is not None else 0
        if features is not None and 0 < table_bytes <= 48 << 30 and psutil.virtual_memory().available > 2 * table_bytes + (32 << 30):
            host = features.cpu().numpy()
            nb2 = max(cores, nb // 4)
            e2, s2, n2 = timed(nb2, cores, host)
            out["with_gather"] = {"edges_per_sec": e2 / s2, "feature_gather_GBps": n2 * host.shape[1] * 4 / s2 / 1e9,
                                  "sample": f"{nb2} batches, sampling + gather of {n2} rows from a host copy of the table, "
                                            f"{s2:.1f} s on {cores} threads"}
            del host
    except Exception as ex:       # reported baseline only: never fail the bench over it
        out["with_gather"] = {"error": repr(ex)[:200]}
    return out


if __name__ == "__main__":
    main()
