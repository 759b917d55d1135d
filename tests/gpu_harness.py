"""Runs a Workload through liblegion_hip.so (GPU) and through the oracle (CPU) side by side."""
import numpy as np
import torch

from legion_amd import engine
from oracle import ffi


class GpuSide:
    def __init__(self, wl, batch_size, fanout, cache_memory=0, feature_rows=None, pipeline_depth=1):
        dev = torch.device("cuda:0")
        self.wl, self.fanout, self.batch_size = wl, list(fanout), batch_size
        self.indptr = torch.from_numpy(wl.indptr).to(dev)
        self.col = torch.from_numpy(wl.col).to(dev)
        self.features = torch.from_numpy(wl.features).to(dev) if wl.features is not None else None
        self.graph = engine.GraphStorage(wl.P, self.indptr, self.col)
        self.feature = engine.FeatureStorage(wl.P, self.features, wl.N, wl.D)
        for (p, mode), (ids, labels) in wl.sets.items():
            self.feature.set_ids(p, mode, ids, labels)
        self.cache = engine.UnifiedCache(cache_memory, wl.D, 1, wl.P, wl.N)
        self.pools = []
        for p in range(wl.P):
            self.cache.init_controller(p)
            pool = engine.MemoryPool(p, wl.N, batch_size, fanout, wl.D, pipeline_depth)
            rows = feature_rows if feature_rows is not None else pool.num_ids
            if wl.D > 0:
                pool.alloc_features(rows)
            self.pools.append(pool)

    def run(self, dev_id, counter, mode, is_presc=False, batch_size=None):
        engine.enqueue_batch(None, self.graph, self.feature, self.cache, self.pools[dev_id],
                             batch_size or self.batch_size, counter, dev_id, mode, is_presc, self.fanout)
        torch.cuda.synchronize()
        out = engine.read_batch(self.pools[dev_id])
        out["cache_search_buffer"] = self.pools[dev_id].buffer("cache_search_buffer")[:max(int(out["node_counter"][1]), 0)].cpu().numpy().copy()
        if is_presc:
            out.pop("float_features", None)      # PreSC runs ops 0,3,6,..,last only: nothing is gathered
        return out

    def close(self):
        for p in self.pools:
            p.close()
        self.cache.close()
        self.feature.close()
        self.graph.close()


class CpuSide:
    def __init__(self, wl, batch_size, fanout, feature_rows=None):
        self.wl, self.fanout, self.batch_size = wl, list(fanout), batch_size
        self.graph = ffi.OracleGraph(wl.P, wl.indptr, wl.col)
        self.pools = [ffi.OraclePool(wl.N, batch_size, fanout,
                                     (feature_rows if feature_rows is not None else ffi.num_ids_for(batch_size, fanout)) if wl.D > 0 else 0,
                                     wl.D) for _ in range(wl.P)]
        self.caches = None                      # list of OracleCache per clique once built
        self.Kg = 1
        self.node_access = [np.zeros(wl.N, dtype=np.uint64) for _ in range(wl.P)]
        self.edge_access = [np.zeros(wl.N, dtype=np.uint64) for _ in range(wl.P)]
        self.max_ids = [0] * wl.P

    def run(self, dev_id, counter, mode, is_presc=False, batch_size=None):
        ids, labels = self.wl.sets[(dev_id, mode)]
        cache = self.caches[dev_id // self.Kg] if (self.caches and not is_presc) else None
        self.pools[dev_id].run_batch(self.graph, cache, self.wl.features, ids, labels,
                                     batch_size or self.batch_size, counter, mode, is_presc,
                                     self.node_access[dev_id] if is_presc else None,
                                     self.edge_access[dev_id] if is_presc else None)
        out = self.pools[dev_id].read_batch()
        if is_presc:
            self.max_ids[dev_id] = max(self.max_ids[dev_id], int(out["node_counter"][7]))
            out.pop("float_features", None)
        return out

    def build_cache(self, cache_agg_mode, cache_memory=None, capacity=None, train_step=1, counters=(0, 0)):
        wl = self.wl
        Kg = min(1 << cache_agg_mode, wl.P)
        self.Kg = Kg
        self.caches = []
        for ki in range(wl.P // Kg):
            c = ffi.OracleCache(wl.N, wl.D, Kg, ki)
            c.candidate_selection(self.node_access[ki * Kg:(ki + 1) * Kg], self.edge_access[ki * Kg:(ki + 1) * Kg])
            if capacity is not None:
                c.set_capacity(*capacity)
            else:
                # the reference passes controller j (not ki*Kg+j), cache.cu:462
                c.cost_model(cache_memory, wl.indptr, counters, self.max_ids[:Kg], train_step)
            c.fill_up(wl.features, wl.indptr, wl.col)
            self.graph.attach_cache(c)
            self.caches.append(c)
        return self.caches

    def close(self):
        for p in self.pools:
            p.close()
        for c in self.caches or []:
            c.close()
