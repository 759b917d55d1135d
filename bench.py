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
`roofline.unique_row_frac` says how many of that launch's rows are distinct (a repeat may be served by the
256 MiB Infinity Cache); `roofline.cold` is the same launch over rows that never repeat (the HBM-only figure),
`roofline.traffic` the FETCH_SIZE + WRITE_SIZE bytes per launch collected by child processes of this run.
`other_shapes[0]` (N = 1, default command) is Legion's default batch size, B = 8000, on the same tables.
`cpu_baseline` is the oracle's C restatement (oracle/, test infrastructure) timed on the host cores
on a bounded sample of the same batches -- a reported baseline, not a target.
The legs around the timed region (argument parsing, the boundary leg, the PMC children, the pieces run_leg strings
together) live in tools/bench_legs.py; this file is the entry point, the leg itself and the CPU baseline.
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

from tools import bench_legs as legs  # noqa: E402
from tools.bench_legs import HBM_PEAK_GBPS  # noqa: E402


def main():
    args = legs.parse_args()
    # the library logs the reference's lines ("Alpha: ...", "Feat capacity: ...") on stdout from every
    # rank; keep the real stdout for the one JSON line and send everything else to stderr
    sys.stdout.flush()
    one_line = legs.OneLine(os.dup(1))
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
    c.one_line = one_line

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
                one_line.refresh()
            if args.striped_replica_memory > 0:
                out_rep = run_leg(c, engine, synth, True, args.striped_replica_memory, headline=False)
                if rank == 0:
                    out["striped_replica"] = out_rep["json"]
                    one_line.refresh()
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

    # ---- N = 1, the default shape: Legion's default batch size on the same tables, driver-timed in the same run ----------------
    # (like the boundary and traffic legs, part of the FULL default run only: --no-boundary, which every measurement script passes, skips it)
    if world == 1 and not args.no_other_shapes and not args.no_boundary and args.placement == "hbm" and B == 1024 and args.nodes == 0 and not args.no_cache:
        out["other_shapes"] = []
        for sb, sf in ((8000, [25, 10]), (8000, [15, 10, 5])):      # Legion's default batch size at BASELINE's two fan-outs
            try:
                c2 = shape_context(c, synth, sb, sf, args.other_shapes_steps, 2)
                if c2 is not None:
                    out["other_shapes"].append(run_leg(c2, engine, synth, False, 0, headline=False, shape_leg=True)["json"])
                    one_line.refresh()
            except Exception as e:      # the headline stands
                import traceback
                traceback.print_exc()
                out["other_shapes"].append({"error": f"{type(e).__name__}: {e}"[:400], "batch": sb, "fanout": sf})

    if rank == 0:
        try:
            if world == 1 and not args.no_boundary and args.placement == "hbm":
                out.update(legs.boundary_leg(args, c.fanout))
                one_line.refresh()
            if args.cpu_seconds > 0 and world == 1:      # reported at N = 1 only
                out["cpu_baseline"] = cpu_baseline(indptr, col, c.mine, N, B, c.fanout, c.n_warm, args.cpu_seconds,
                                                   features if args.placement == "hbm" else None)
                one_line.refresh()
            if world == 1 and not args.no_boundary and not args.no_traffic_leg and args.placement == "hbm":
                legs.measured_traffic(args, out["roofline"], c.G)
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
    if one_line.needs_hard_exit:      # armed with the ctypes stand-in (no C compiler for tools/exit_line.c): a Python callable
        sys.stderr.flush()            # registered with libc must not be called after the interpreter is gone
        os._exit(0)


def fail_as_requested(how):
    """--fail-extra-leg: end (or stall) this process the way a leg behind the headline might."""
    if how == "raise":
        raise RuntimeError("requested by --fail-extra-leg")
    if how == "exit":
        ctypes.CDLL(None).exit(3)
    if how == "sigterm":
        os.kill(os.getpid(), signal.SIGTERM)
    while True:
        ctypes.CDLL(None).sleep(5)      # (a main thread that never comes back from a C call)


def run_leg(c, engine, synth, stripe, replica_memory, headline, bulk=False, shape_leg=False):
    """One cache layout over the resident workload: objects -> PreSC -> hotness all-reduce -> cost model -> fills -> pipeline ->
    counting pass -> warm-up -> timed regions -> eager pass with HIP events around the gathers (-> headline: the cold-row regather).
    Returns {"json": rank 0's report of the leg}.  The headline leg also verifies; the extra legs are shorter.  shape_leg: another
    batch shape of the same workload on one GPU (`other_shapes`), reported in the extra legs' short form."""
    args, world, rank, dev, use_dist = c.args, c.world, c.rank, c.dev, c.use_dist
    if not headline and not shape_leg and args.fail_extra_leg and rank == 0:        # (tests of OneLine)
        fail_as_requested(args.fail_extra_leg)
    fanout, H, N, D, B, G = c.fanout, c.H, c.N, c.D, c.B, c.G
    hy_cpu = hy_gpu = 0
    n_timed, mine = c.n_timed, c.mine
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
    collective = legs.hotness_collective(c, engine, cache, d, red_dev) if use_dist else None
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
    elif args.hybrid:
        hy_cpu, hy_gpu = (int(x) for x in args.hybrid.split(","))
        cache.hybrid_init(feature, graph, hy_cpu, hy_gpu)                   # every GPU from its own counters: no collective result is used
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
        pipe = legs.BulkPipe(engine.Pipeline(graph, feature, cache, d, B, fanout, G, feature_rows, False, args.slots, arena="shared"), use_dist)
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

    # ---- counting pass, timed regions, the verification of what the last replay left, the profiled pass ---------------------------
    have_map = cache.node_capacity(d) > 0 and not args.no_cache
    node_map = cache.array("node_map", d) if have_map else torch.empty(0, dtype=torch.int32, device=dev)
    counted = legs.counting_pass(c, synth, pipe, cache, d, node_map, feature_rows, headline or shape_leg, stripe, bulk)
    edges, rows, hop_edges, hop_slots = counted.edges, counted.rows, counted.hop_edges, counted.hop_slots
    timed = legs.timed_regions(c, engine, pipe, d, headline)
    elapsed_max = elapsed = timed.elapsed_max                        # (`elapsed`: the rank-0 sampler / gather split below)
    if headline and not args.no_verify and timed.last_group is not None:
        legs.verify_last_group(c, synth, pipe, counted, timed.last_group)
    prof, elapsed_profiled = legs.profiled_pass(c, pipe)
    state_bytes, lds_buckets = pipe.pools[0][0].state_bytes(), pipe.pools[0][0].lds_buckets()

    # ---- roofline of the dominant kernel: the last hop's gather (op 3H+1) -------------------------
    last_op = 3 * H + 1
    t_last, n_last = prof.get(last_op, (0.0, 0))
    t_last *= 1e-3
    rows_last = int(rows[:, H].sum())
    bytes_per_row = 8 * D + 8
    achieved = rows_last * bytes_per_row / t_last / 1e9 if t_last > 0 else 0.0
    t_all_gathers = sum(v[0] for v in prof.values()) * 1e-3
    payload_gbps = float(rows.sum() * D * 4) / t_all_gathers / 1e9 if t_all_gathers > 0 else 0.0
    tot_edges = torch.tensor([float(edges.sum())], dtype=torch.float64, device=dev)
    gather_bytes_t = torch.tensor([float(rows.sum() * D * 4)], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(tot_edges)
        dist.all_reduce(gather_bytes_t)
    # sampler side: bytes = sum_h [S_h*25/f_h + E_h*44 + U_h*8]; time = step time minus the gathers
    samp_bytes = sum(float(hop_slots[:, h].sum()) * 25.0 / fanout[h] + float(hop_edges[:, h].sum()) * 44.0 +
                     float(rows[:, h + 1].sum()) * 8.0 for h in range(H))
    t_sampling = max(elapsed - t_all_gathers, 1e-9)
    per_rank = legs.per_rank_info(c, timed, counted, achieved, t_last, n_last, pipe, bulk) if use_dist else None
    traffic_committed, traffic_committed_src, rocprof_us, rocprof_src = legs.committed_profiles(c, rows_last, n_last)

    out = None
    if rank == 0:
        here = locals()                      # (the report is assembled in tools/bench_legs.py from exactly the values it names)
        out = legs.leg_report(types.SimpleNamespace(**{k: here[k] for k in legs.REPORT_NAMES}))
    # ---- the same launch alone and over rows that never repeat (a one-GPU run: the figure is about the kernel, not the job).  The
    #      headline's line exists by now: from here on it goes out even if an optional leg ends the process (a native exit() of the
    #      library, SIGTERM, a hang) -- this leg, the other shapes, the boundary, the CPU baseline and the traffic children follow. ----
    if headline and world == 1 and rank == 0 and getattr(c, "one_line", None) is not None:
        c.one_line.arm(out, args.post_legs_deadline)
    if (headline or shape_leg) and world == 1 and not bulk and not args.no_cold_leg and (not args.no_boundary or args.cold_leg):
        try:
            if headline and args.fail_extra_leg:             # (tests of the guard at N = 1: the first leg behind the headline fails this way)
                fail_as_requested(args.fail_extra_leg)
            cold = legs.cold_regather(c, pipe, node_map, bytes_per_row)
            out["roofline"].update({"cold": cold.get("cold"), "alone": cold.get("alone"), "cold_note": cold.get("note")})
        except Exception as e:      # a diagnostic: never the reason a headline is lost
            out["roofline"].update({"cold": None, "alone": None, "cold_note": repr(e)[:300]})
        if headline and getattr(c, "one_line", None) is not None:
            c.one_line.refresh()
    pipe.close()
    # this leg's objects go before the next leg builds its own (other logical-GPU numbering, other cache layout)
    cache.close()
    feature.close()
    graph.close()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    return {"json": out}


def shape_context(c, synth, batch, fanout, steps, warmup):
    """The headline's context with another batch size / fan-out: group size by the same rule (halved while the lanes in flight would not
    fit 0.7 of the free HBM), K = steps, seeds for exactly that run."""
    c2 = types.SimpleNamespace(**vars(c))
    c2.args = types.SimpleNamespace(**vars(c.args))
    c2.args.steps, c2.args.warmup, c2.args.batch, c2.args.fanout = steps, warmup, batch, ",".join(str(f) for f in fanout)
    c2.B, c2.fanout, c2.H = batch, list(fanout), len(fanout)
    G = 1
    while G * 2 <= 512 and G * 2 * batch <= 524288:
        G *= 2
    num_ids, per = batch, batch
    for f in fanout:
        per *= f
        num_ids += per
    lane_bytes = num_ids * 56 + per * 28 + (num_ids // (8 if len(fanout) <= 2 else 16)) * c.D * 4
    while G > 1 and G * c.args.slots * lane_bytes > torch.cuda.mem_get_info(c.dev)[0] * 7 // 10:
        G //= 2
    c2.G = G
    c2.n_warm, c2.n_timed = warmup * G, steps * G
    need = max((c2.n_warm + c2.n_timed + 2) * batch + batch, (c.args.presc_steps + 2) * batch)
    all_seeds = synth.seed_ids(c.N, min(max(need * 2, c.N // 10), c.N), 11)
    c2.mine = np.ascontiguousarray(all_seeds)
    c2.epoch_batches = ((c2.mine.size - 1) // batch) // G * G
    if c2.epoch_batches < G:
        return None
    c2.wrap = c2.epoch_batches if c2.n_warm + c2.n_timed > c2.epoch_batches else None
    return c2


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
