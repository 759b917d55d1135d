"""How often the SAME feature row is gathered by several mini-batches of one launch group (RMAT-26, B = 1024, [25,10],
256 lanes): unique rows / rows over the group, and the share of the group's rows that the hottest k vertices account for.
Decides whether ordering the gather for cross-lane reuse could save HBM reads.   python tools/group_row_reuse.py [--scale 26]"""
import argparse, json, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from legion_amd import engine, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=int, default=26)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--group", type=int, default=256)
    ap.add_argument("--fanout", type=str, default="25,10")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    fanout = [int(x) for x in a.fanout.split(",")]
    N, D, B, G = 1 << a.scale, 4, a.batch, a.group
    indptr, col = synth.rmat_csr_device(a.scale, 16, 20231, dev)
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
    pipe = engine.Pipeline(graph, feature, cache, 0, B, fanout, G, pool.num_ids, True, 1)
    sl = pipe.submit(G)
    pipe.wait(sl)
    ids = []
    for lane in range(G):
        pl = pipe.pools[sl][lane]
        nc = pl.buffer("node_counter").cpu().numpy()
        n = int(nc[9 + len(fanout)])
        ids.append(pl.buffer("sampled_ids")[:n].clone())
    allids = torch.cat(ids)
    uniq, cnt = torch.unique(allids, return_counts=True)
    cnt_sorted, _ = torch.sort(cnt, descending=True)
    cum = torch.cumsum(cnt_sorted, 0).double() / allids.numel()
    out = {"rows": int(allids.numel()), "unique_rows": int(uniq.numel()), "unique_over_rows": uniq.numel() / allids.numel()}
    for k in (1 << 10, 1 << 14, 1 << 16, 1 << 18, 1 << 20):
        if k <= cnt_sorted.numel():
            out[f"share_of_hottest_{k}"] = float(cum[k - 1])
    out["rows_gathered_once"] = int((cnt == 1).sum())
    print(json.dumps(out))


if __name__ == "__main__":
    main()
