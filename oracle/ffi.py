"""ctypes binding of the CPU oracle (oracle/legion_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg.  Nothing under legion_amd/ imports this module.
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# LEGION_ORACLE_LIB=<path>: another build of the same library (oracle/_build/san/: the sanitizer leg, tests/test_sanitizers_cpu.py)
LIB_PATH = os.environ.get("LEGION_ORACLE_LIB") or os.path.join(HERE, "_build", "liblegion_oracle.so")
MAX_DEVICE = 8
INTRABATCH_CON = 3

c_i32, c_i64, c_u64, c_p = ctypes.c_int32, ctypes.c_int64, ctypes.c_uint64, ctypes.c_void_p
P_I32 = ctypes.POINTER(ctypes.c_int32)
P_I8 = ctypes.POINTER(ctypes.c_int8)
P_U32 = ctypes.POINTER(ctypes.c_uint32)
P_U64 = ctypes.POINTER(ctypes.c_uint64)
P_I64 = ctypes.POINTER(ctypes.c_int64)
P_F32 = ctypes.POINTER(ctypes.c_float)


class Pool(ctypes.Structure):
    _fields_ = [("total_num_nodes", c_i32), ("num_ids", c_i32),
                ("sampled_ids", P_I32), ("labels", P_I32), ("agg_src_ids", P_I32), ("agg_dst_ids", P_I32),
                ("agg_src_off", P_I32), ("agg_dst_off", P_I32), ("cache_search_buffer", P_I32),
                ("tmp_part_ind", P_I8), ("tmp_part_off", P_I32), ("accessed_map", P_U32),
                ("position_map", P_I32), ("node_counter", c_i32 * 16), ("edge_counter", c_i32 * 16),
                ("float_features", P_F32), ("feature_rows", c_i64), ("feature_dim", c_i32)]


class Graph(ctypes.Structure):
    _fields_ = [("partition_count", c_i32),
                ("csr_node_index", P_I64 * (MAX_DEVICE + 1)),
                ("csr_dst_node_ids", P_I32 * (MAX_DEVICE + 1))]


class Cache(ctypes.Structure):
    _fields_ = [("total_num_nodes", c_i32), ("feature_dim", c_i32), ("Kg", c_i32), ("Ki", c_i32),
                ("node_capacity", c_i32), ("edge_capacity", c_i32),
                ("QF", P_I32), ("QT", P_I32), ("AF", P_U64), ("AT", P_U64),
                ("node_map", P_I32), ("edge_index_map", P_I8), ("edge_offset_map", P_I32),
                ("feat_cache", P_F32 * MAX_DEVICE), ("topo_indptr", P_I64 * MAX_DEVICE),
                ("topo_col", P_I32 * MAX_DEVICE),
                ("hybrid", c_i32), ("cpu_cache_capacity", c_i32), ("gpu_cache_capacity", c_i32), ("cpu_cache", P_F32)]


class Steps(ctypes.Structure):
    _fields_ = [("train_step", c_i32), ("valid_step", c_i32), ("test_step", c_i32), ("epoch", c_i32),
                ("raw_batch_size", c_i32), ("train_bs", c_i32 * MAX_DEVICE), ("valid_bs", c_i32 * MAX_DEVICE),
                ("test_bs", c_i32 * MAX_DEVICE)]


_lib = None


REF_GRAPH_CACHE = os.path.join(HERE, "_ref", "ref_graph_cache")     # the reference's own GraphCache kernels behind this repo's driver (Makefile target `ref`)


def build():
    subprocess.check_call(["make", "-s", "-C", HERE])
    subprocess.check_call(["make", "-s", "-C", HERE, "ref"])       # (a no-op where /root/reference does not exist: the GPU box uses the prebuilt binary)
    return LIB_PATH


def load():
    global _lib
    if _lib is not None:
        return _lib
    srcs = [os.path.join(HERE, f) for f in ("legion_oracle.c", "dgl_semantics.c", "legion_oracle.h")]
    if "LEGION_ORACLE_LIB" not in os.environ and (not os.path.exists(LIB_PATH) or any(os.path.getmtime(f) > os.path.getmtime(LIB_PATH) for f in srcs)):
        build()
    L = ctypes.CDLL(LIB_PATH)
    PP, PG, PC = ctypes.POINTER(Pool), ctypes.POINTER(Graph), ctypes.POINTER(Cache)
    sig = {
        "lgo_minstd_pow": (ctypes.c_uint32, [c_u64]),
        "lgo_draw": (c_i32, [c_i32, c_i32]),
        "lgo_pool_create": (PP, [c_i32, c_i32, c_i32, c_i64, c_i32]),
        "lgo_pool_destroy": (None, [PP]),
        "lgo_batch_generate": (None, [PP, P_I32, P_I32, c_i32, c_i32, c_i32, c_i32]),
        "lgo_counter_update": (None, [P_I32, P_I32, c_i32, c_i32, c_i32]),
        "lgo_random_sample": (None, [PP, PG, c_i32, c_i32, P_I8, P_I32, ctypes.c_int, P_U64]),
        "lgo_construct_graph": (None, [PP]),
        "lgo_clear_pos_map": (None, [PP]),
        "lgo_hotness_measure": (None, [PP, P_U64]),
        "lgo_cache_create": (PC, [c_i32, c_i32, c_i32, c_i32]),
        "lgo_cache_destroy": (None, [PC]),
        "lgo_candidate_selection": (None, [PC, ctypes.POINTER(P_U64), ctypes.POINTER(P_U64)]),
        "lgo_cost_model": (c_i32, [PC, c_i64, P_I64, P_U64, P_I32, c_i32, P_F32]),
        "lgo_fill_up": (None, [PC, P_F32, P_I64, P_I32]),
        "lgo_hybrid_init": (None, [PC, P_U64, P_F32, c_i32, c_i32]),
        "lgo_find_topo": (None, [PC, P_I32, P_I8, P_I32, c_i32]),
        "lgo_find_feat": (None, [PC, PP, c_i32]),
        "lgo_feature_cache_lookup": (None, [PC, PP, P_F32, c_i32]),
        "lgo_run_batch": (c_i64, [PP, PG, PC, P_F32, P_I32, P_I32, c_i32, c_i32, c_i32, P_I32, c_i32, c_i32,
                                  ctypes.c_int, P_U64, P_U64]),
        "lgo_coordinate": (None, [ctypes.POINTER(Steps), c_i32, P_I32, P_I32, P_I32, c_i32, c_i32]),
        "lgo_max_step": (c_i32, [ctypes.POINTER(Steps)]),
        "lgo_current_mode": (c_i32, [ctypes.POINTER(Steps), c_i32]),
        "lgo_local_batch_id": (c_i32, [ctypes.POINTER(Steps), c_i32]),
        "lgo_current_batchsize": (c_i32, [ctypes.POINTER(Steps), c_i32, c_i32]),
        "lgo_bench_batches": (c_i64, [PG, c_i32, P_I32, c_i32, c_i32, P_I32, c_i32, c_i32, c_i32, c_i32, P_F32,
                                      c_i32, ctypes.POINTER(ctypes.c_double), P_I64]),
        # dgl_semantics.c: the DGL-semantics throughput baseline (no parity role)
        "lgo_dgl_sample_batch": (c_i64, [P_I64, P_I32, P_I32, c_i32, P_I32, c_i32, c_u64, P_I64, P_I32, P_I32, P_I32, P_I32]),
        "lgo_dgl_bench_batches": (c_i64, [P_I64, P_I32, P_I32, c_i32, c_i32, P_I32, c_i32, c_i32, c_i32, c_i32,
                                          ctypes.POINTER(ctypes.c_double), P_I64]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
    _lib = L
    return L


def _p(a, ptype):
    return a.ctypes.data_as(ptype) if a is not None else ctypes.cast(None, ptype)


def num_ids_for(batch_size, fanout):
    n, per = batch_size, batch_size
    for f in fanout:
        per *= f
        n += per
    return n


class OracleGraph:
    """Full CSR in slot P; cached CSRs are attached from an OracleCache."""

    def __init__(self, partition_count, indptr, col):
        self.indptr = np.ascontiguousarray(indptr, dtype=np.int64)
        self.col = np.ascontiguousarray(col, dtype=np.int32)
        self.g = Graph()
        self.g.partition_count = partition_count
        self.g.csr_node_index[partition_count] = _p(self.indptr, P_I64)
        self.g.csr_dst_node_ids[partition_count] = _p(self.col, P_I32)

    def attach_cache(self, cache):
        c = cache.c.contents
        for j in range(c.Kg):
            dev = c.Ki * c.Kg + j
            self.g.csr_node_index[dev] = c.topo_indptr[j]
            self.g.csr_dst_node_ids[dev] = c.topo_col[j]


class OracleCache:
    def __init__(self, total_num_nodes, feature_dim, Kg=1, Ki=0):
        self.L = load()
        self.N, self.D, self.Kg = total_num_nodes, feature_dim, Kg
        self.c = self.L.lgo_cache_create(total_num_nodes, feature_dim, Kg, Ki)

    def candidate_selection(self, node_access_list, edge_access_list):
        self._na = [np.ascontiguousarray(a, dtype=np.uint64) for a in node_access_list]
        self._ea = [np.ascontiguousarray(a, dtype=np.uint64) for a in edge_access_list]
        na = (P_U64 * self.Kg)(*[_p(a, P_U64) for a in self._na])
        ea = (P_U64 * self.Kg)(*[_p(a, P_U64) for a in self._ea])
        self.L.lgo_candidate_selection(self.c, na, ea)

    def cost_model(self, cache_memory, indptr, counters, max_id_num, train_step):
        cnt = (c_u64 * 2)(int(counters[0]), int(counters[1]))
        mid = np.ascontiguousarray(max_id_num, dtype=np.int32)
        trans = ctypes.c_float(0)
        alpha = self.L.lgo_cost_model(self.c, int(cache_memory), _p(indptr, P_I64), cnt, _p(mid, P_I32),
                                      int(train_step), ctypes.byref(trans))
        return alpha, trans.value

    def set_capacity(self, node_capacity, edge_capacity):
        self.c.contents.node_capacity = int(node_capacity)
        self.c.contents.edge_capacity = int(edge_capacity)

    def fill_up(self, features, indptr, col):
        self.L.lgo_fill_up(self.c, _p(features, P_F32), _p(indptr, P_I64), _p(col, P_I32))

    def hybrid_init(self, node_access, features, cpu_cache_capacity, gpu_cache_capacity):
        """The hybrid CPU-cache / GPU-cache tier of one GPU from its own hotness counters (cache.cu:614-670)."""
        self._na = [np.ascontiguousarray(node_access, dtype=np.uint64)]
        self.L.lgo_hybrid_init(self.c, _p(self._na[0], P_U64), _p(features, P_F32), int(cpu_cache_capacity),
                               int(gpu_cache_capacity))

    def arr(self, name, dtype):
        return np.ctypeslib.as_array(getattr(self.c.contents, name), shape=(self.N,)).astype(dtype, copy=True)

    @property
    def node_capacity(self):
        return int(self.c.contents.node_capacity)

    @property
    def edge_capacity(self):
        return int(self.c.contents.edge_capacity)

    def close(self):
        if self.c:
            self.L.lgo_cache_destroy(self.c)
            self.c = None


class OraclePool:
    def __init__(self, total_num_nodes, batch_size, fanout, feature_rows=0, feature_dim=0):
        self.L = load()
        self.fanout = [int(f) for f in fanout]
        self.num_ids = num_ids_for(batch_size, self.fanout)
        self.D = feature_dim
        self.p = self.L.lgo_pool_create(total_num_nodes, self.num_ids, batch_size, int(feature_rows), feature_dim)

    def run_batch(self, graph, cache, features, all_ids, all_labels, batch_size, counter, mode, is_presc,
                  node_access=None, edge_access=None):
        ids = np.ascontiguousarray(all_ids, dtype=np.int32)
        labels = np.ascontiguousarray(all_labels if all_labels is not None else np.zeros_like(ids), dtype=np.int32)
        fan = np.asarray(self.fanout, dtype=np.int32)
        return self.L.lgo_run_batch(self.p, ctypes.byref(graph.g), cache.c if cache is not None else None,
                                    _p(features, P_F32), _p(ids, P_I32), _p(labels, P_I32), int(ids.size),
                                    int(batch_size), int(counter), _p(fan, P_I32), len(self.fanout), int(mode),
                                    1 if is_presc else 0, _p(node_access, P_U64), _p(edge_access, P_U64))

    def read_batch(self):
        p = self.p.contents
        nc = np.array(p.node_counter[:], dtype=np.int32)
        ec = np.array(p.edge_counter[:], dtype=np.int32)
        hop_num = int(nc[INTRABATCH_CON * 3 - 1])
        n_nodes = max(int(nc[INTRABATCH_CON * 3 + hop_num]), 0)
        n_edges = max(int(ec[INTRABATCH_CON * 3 + hop_num]), 0)

        def take(ptr, n, dtype):
            if n == 0:
                return np.zeros(0, dtype=dtype)
            return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True)

        out = {"node_counter": nc, "edge_counter": ec, "hop_num": hop_num,
               "sampled_ids": take(p.sampled_ids, n_nodes, np.int32),
               "labels": take(p.labels, max(int(nc[INTRABATCH_CON * 3]), 0), np.int32),
               "agg_src_off": take(p.agg_src_off, n_edges, np.int32),
               "agg_dst_off": take(p.agg_dst_off, n_edges, np.int32),
               "agg_src_ids": take(p.agg_src_ids, n_edges, np.int32),
               "agg_dst_ids": take(p.agg_dst_ids, n_edges, np.int32),
               "cache_search_buffer": take(p.cache_search_buffer, int(nc[1]) if nc[1] > 0 else 0, np.int32)}
        if p.feature_rows > 0 and self.D > 0 and n_nodes > 0:
            out["float_features"] = np.ctypeslib.as_array(p.float_features, shape=(n_nodes * self.D,)) \
                .reshape(n_nodes, self.D).copy()
        return out

    def close(self):
        if self.p:
            self.L.lgo_pool_destroy(self.p)
            self.p = None
