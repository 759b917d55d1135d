"""Scratch probe: eager multi-lane throughput (L pools, L streams) at a given batch size."""
import argparse, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from legion_amd import engine, synth

ap = argparse.ArgumentParser()
ap.add_argument("--scale", type=int, default=26); ap.add_argument("--batch", type=int, default=1024)
ap.add_argument("--lanes", type=str, default="1,2,4,8"); ap.add_argument("--steps", type=int, default=200)
a = ap.parse_args()
dev = torch.device("cuda:0"); N = 1 << a.scale; D = 128; B = a.batch; fan = [25, 10]
indptr, col = synth.rmat_csr_device(a.scale, 16, 20231, dev)
feats = synth.features_device(N, D, 7, dev)
seeds = synth.seed_ids(N, N // 10, 11)
graph = engine.GraphStorage(1, indptr, col); feature = engine.FeatureStorage(1, feats)
feature.set_ids(0, 0, seeds, None)
cache = engine.UnifiedCache(8 << 30, D, 64, 1, N); cache.init_controller(0)
pool0 = engine.MemoryPool(0, N, B, fan, D)
for it in range(64):
    engine.enqueue_batch(None, graph, feature, cache, pool0, B, it, 0, 0, True, fan)
torch.cuda.synchronize()
cache.candidate_selection(0, graph); cache.cost_model(feature, graph, (0, 0), 64); cache.fill_up(feature, graph)
rows = int(cache.max_id_num(0) * 1.2)
for L in [int(x) for x in a.lanes.split(",")]:
    pools = [engine.MemoryPool(0, N, B, fan, D) for _ in range(L)]
    for p in pools: p.alloc_features(rows)
    streams = [torch.cuda.Stream() for _ in range(L)]
    def go(k0, n):
        for k in range(n):
            engine.enqueue_batch(streams[k % L], graph, feature, cache, pools[k % L], B, k0 + k, 0, 0, False, fan)
    go(0, 20); torch.cuda.synchronize()
    t0 = time.perf_counter(); go(20, a.steps); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    # host-only cost: time the enqueue loop without waiting
    t1 = time.perf_counter(); go(20, a.steps); th = time.perf_counter() - t1; torch.cuda.synchronize()
    ec = pools[0].buffer("edge_counter").cpu().numpy(); nc = pools[0].buffer("node_counter").cpu().numpy()
    print(f"B={B} lanes={L}: {dt / a.steps * 1e6:.1f} us/batch (host enqueue {th / a.steps * 1e6:.1f} us/batch); last batch edges={ec[11]} nodes={nc[11]}", flush=True)
    for p in pools: p.close()
