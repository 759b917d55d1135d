"""how many DISTINCT 128-byte lines of the column array do the last hop's picks of one launch group touch, against
the lines sample_kernel fetches (FETCH_SIZE of its hop-2 launch / 128)?  If distinct / fetched is near 1 the kernel's 13.7 x traffic
over its algorithmic bytes is the part's floor for this draw (one 128-byte request per 8-byte pick); if it is well below 1 the same line
is fetched again by another lane of the group and an issue order that groups a source row's picks could save reads.

The picks are recomputed from what a group's batches hold -- hop 2's frontier = the edges of hop 1 in slot order, slot idx = 10 i + k picks
column start(src) + draw(idx, deg(src)) (SS/engine/operator_impl.cu:235-243; the draw by the library's own legion_draw_batch) -- and
checked against the edges the sampler produced.

    python tools/pick_line_reuse.py [--scale 26] [--batch 1024] [--group 512] [--fanout 25,10] [--fetched-mb 1603.3] [--entry-bytes 8]
        -> one JSON line (+ a markdown paragraph on stderr)"""
import argparse
import ctypes
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from legion_amd import engine, lib, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=int, default=26)
    ap.add_argument("--edge-factor", type=int, default=16)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--group", type=int, default=512)
    ap.add_argument("--fanout", type=str, default="25,10")
    ap.add_argument("--entry-bytes", type=int, default=8, help="8: the {neighbour, cache slot} pairs of the column slots (the headline); 4: the plain column array")
    ap.add_argument("--fetched-mb", type=float, default=0.0, help="HBM bytes read by one launch of the last hop's sample_kernel over such a group (PMC, MB = 1e6)")
    ap.add_argument("--other-mb", type=float, default=0.0,
                    help="of those, what the kernel streams besides its picks (row headers, frontier ids): subtracted before the comparison")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    L = lib.load()
    fanout = [int(x) for x in a.fanout.split(",")]
    H, f_last = len(fanout), fanout[-1]
    N, D, B, G = 1 << a.scale, 4, a.batch, a.group
    indptr, col = synth.rmat_csr_device(a.scale, a.edge_factor, 20231, dev)
    torch.cuda.empty_cache()
    feats = synth.features_device(N, D, 7, dev)
    seeds = synth.seed_ids(N, max(N // 10, 4 * G * B), 11)
    graph = engine.GraphStorage(1, indptr, col)
    feature = engine.FeatureStorage(1, feats)
    feature.set_ids(0, engine.TRAINMODE, seeds, None)
    cache = engine.UnifiedCache(1 << 20, D, 1, 1, N)
    cache.init_controller(0)
    pool = engine.MemoryPool(0, N, B, fanout, D, pipeline_depth=1)
    engine.enqueue_batch(None, graph, feature, cache, pool, B, 0, 0, engine.TRAINMODE, True, fanout)
    torch.cuda.synchronize()
    cache.candidate_selection(0, graph)
    cache.set_capacity(16, 16)
    cache.fill_up(feature, graph)
    pipe = engine.Pipeline(graph, feature, cache, 0, B, fanout, G, pool.num_ids, False, 1)
    sl = pipe.submit(G)                 # the group the bench's first timed step serves at --warmup 1 (any full group does)
    pipe.wait(sl)
    deg_all = indptr[1:] - indptr[:-1]
    per_line = 128 // a.entry_bytes
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    lines_all, picks, slots, in_lane_distinct, row_lines = [], 0, 0, 0, 0
    for lane in range(G):
        pl = pipe.pools[sl][lane]
        ec = pl.buffer("edge_counter").cpu().numpy()
        nc = pl.buffer("node_counter").cpu().numpy()
        e_lo, e_hi, e_end = int(ec[9 + H - 2]) if H > 1 else 0, int(ec[9 + H - 1]), int(ec[9 + H])
        frontier = (pl.buffer("agg_src_ids")[e_lo:e_hi] if H > 1 else pl.buffer("sampled_ids")[:int(nc[9])]).long()
        n_slots = int(frontier.numel()) * f_last
        idx = torch.arange(n_slots, device=dev, dtype=torch.int32)
        src = frontier[(idx // f_last).long()]
        deg = deg_all[src].to(torch.int32)
        valid = (idx % f_last) < deg                                      # operator_impl.cu:231: k >= deg -> no edge
        pick = torch.empty(n_slots, dtype=torch.int32, device=dev)
        L.legion_draw_batch(st, ctypes.c_void_p(idx.data_ptr()), ctypes.c_void_p(deg.clamp(min=1).data_ptr()), ctypes.c_void_p(pick.data_ptr()), n_slots)
        at = (indptr[src] + pick.long())[valid]
        got = pl.buffer("agg_src_ids")[e_hi:e_end]
        assert at.numel() == got.numel() and bool((col[at] == got).all()), f"lane {lane}: the recomputed picks are not the sampler's"
        ln = at // per_line
        lines_all.append(ln)
        picks += int(at.numel())
        slots += n_slots
        in_lane_distinct += int(torch.unique(ln).numel())
        # the picks of ONE source row fall into deg * entry_bytes / 128 lines: what grouping a row's picks across its slots could merge
        row_lines += int(torch.unique((src[valid] << 30) | ln).numel())          # (vertex < 2^30, line < 2^30: one int64 key)
    all_lines = torch.cat(lines_all)
    uniq, cnt = torch.unique(all_lines, return_counts=True)
    distinct = int(uniq.numel())
    out = {"graph": f"RMAT-{a.scale} EF{a.edge_factor}", "batch": B, "fanout": fanout, "lanes": G, "entry_bytes": a.entry_bytes,
           "last_hop_slots": slots, "last_hop_picks": picks,
           "distinct_lines_group": distinct, "distinct_lines_summed_per_lane": in_lane_distinct,
           "distinct_(lane,source row,line)": row_lines,
           "lines_touched_once": int((cnt == 1).sum()), "picks_per_distinct_line": picks / max(distinct, 1),
           "distinct_lines_bytes": distinct * 128, "picks_bytes": picks * a.entry_bytes}
    if a.fetched_mb > 0:
        fetched = (a.fetched_mb - a.other_mb) * 1e6 / 128
        out.update({"fetched_lines_pmc": fetched, "distinct_over_fetched": distinct / fetched,
                    "per_lane_distinct_over_fetched": in_lane_distinct / fetched, "picks_over_fetched": picks / fetched})
    print(json.dumps(out))
    pipe.close()


if __name__ == "__main__":
    main()
