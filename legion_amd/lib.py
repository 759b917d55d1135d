"""ctypes binding of liblegion_hip.so (include/legion_hip.h).

The library is the product; this module only declares its C ABI to Python.  There is no fallback:
if the shared object is missing or does not load, importing callers get an ImportError that says
how to build it, and nothing else in the package will run.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# LEGION_HIP_LIB=<path>: load another build of the same library (tuning sweeps with other compile-time constants,
# tools/lds_tuning/); the library in place is never replaced
LIB_PATH = os.environ.get("LEGION_HIP_LIB") or os.path.join(_HERE, "liblegion_hip.so")

c_i32, c_i64, c_u64 = ctypes.c_int32, ctypes.c_int64, ctypes.c_uint64
c_p, c_bool = ctypes.c_void_p, ctypes.c_bool
P_I32 = ctypes.POINTER(ctypes.c_int32)
P_U64 = ctypes.POINTER(ctypes.c_uint64)

# name -> (restype, argtypes); the order and meaning follow include/legion_hip.h
SIGNATURES = {
    # 1. operators (SS/engine/operator_impl.cuh:11-63)
    "BatchGenerate": (None, [c_p, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_i32, c_bool, c_i32]),
    "RandomSample": (None, [c_p, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_bool]),
    "FeatureCacheLookup": (None, [c_p, c_p, c_p, c_i32, c_i32]),
    "IOSubmit": (None, [c_p, c_p, c_p, c_i32, c_i32]),
    "IOComplete": (None, [c_p, c_p, c_p, c_i32, c_i32]),
    "legion_enqueue_batch": (None, [c_p, c_p, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_bool, P_I32, c_i32]),
    # 2. objects
    "legion_graph_create": (c_p, [c_i32, c_i32, c_i64, c_p, c_p]),
    "legion_graph_destroy": (None, [c_p]),
    "legion_graph_column_slots": (c_i32, [c_p, c_i32]),
    "legion_graph_cached_csr": (None, [c_p, c_i32, ctypes.POINTER(c_p), ctypes.POINTER(c_p)]),
    "legion_feature_create": (c_p, [c_i32, c_i32, c_i32, c_p]),
    "legion_feature_set_ids": (None, [c_p, c_i32, c_i32, c_p, c_p, c_i32]),
    "legion_feature_destroy": (None, [c_p]),
    "legion_pool_create": (c_p, [c_i32, c_i32, c_i32, P_I32, c_i32, c_i32, c_i32]),
    "legion_pool_alloc_features": (None, [c_p, c_i64]),
    "legion_pool_set_current_pipe": (None, [c_p, c_i32]),
    "legion_pool_set_mode_iter": (None, [c_p, c_i32, c_i32]),
    "legion_pool_num_ids": (c_i32, [c_p]),
    "legion_pool_buffer": (c_p, [c_p, c_i32]),
    "legion_pool_destroy": (None, [c_p]),
    "legion_pool_lds_buckets": (c_i32, [c_p]),
    "legion_pool_state_bytes": (c_i64, [c_p]),
    "legion_pool_error": (c_i32, [c_p]),
    "legion_cache_create": (c_p, [c_i64, c_i32, c_i32, c_i32, c_i32]),
    "legion_cache_init_controller": (None, [c_p, c_i32]),
    "legion_cache_set_replica_memory": (None, [c_p, c_i64]),
    "legion_cache_replica_rows": (c_i32, [c_p, c_i32]),
    "legion_cache_gather_stats": (None, [c_p, c_i32, P_U64]),
    "legion_cache_candidate_selection": (None, [c_p, c_i32, c_p, c_i32]),
    "legion_cache_cost_model": (None, [c_p, c_p, c_p, P_U64, c_i32]),
    "legion_cache_set_capacity": (None, [c_p, c_i32, c_i32]),
    "legion_cache_fill_up": (None, [c_p, c_p, c_p]),
    "legion_cache_hybrid_init": (None, [c_p, c_p, c_p, c_i32, c_i32, c_i32]),
    "legion_cache_hybrid_cpu_cache": (c_p, [c_p, c_i32]),
    "legion_cache_feature_cache": (c_p, [c_p, c_i32]),
    "legion_cache_destroy": (None, [c_p]),
    "legion_set_local_device": (None, [c_i32]),
    "legion_cache_set_peer_max_ids": (None, [c_p, P_I32, c_i32]),
    "legion_cache_fill_up_local": (None, [c_p, c_p, c_p]),
    "legion_cache_export": (None, [c_p, c_p, c_i32, c_p]),
    "legion_cache_import_peer": (None, [c_p, c_p, c_i32, c_i32, c_p]),
    "legion_cache_fill_up_link": (None, [c_p, c_p, c_p]),
    "legion_collective_unique_id": (c_i32, [c_p]),
    "legion_collective_init_rank": (c_i32, [c_p, c_i32, c_i32, c_i32]),
    "legion_collective_allreduce_u64": (c_i32, [c_p, c_i64, ctypes.POINTER(ctypes.c_double)]),
    "legion_collective_destroy": (None, []),
    "legion_cache_allreduce_hotness": (c_i32, [c_p, c_i32, ctypes.POINTER(ctypes.c_double)]),
    "legion_cache_hotness_reduce_path": (c_i32, [c_p, c_i32]),
    "legion_cache_node_capacity": (c_i32, [c_p, c_i32]),
    "legion_cache_edge_capacity": (c_i32, [c_p, c_i32]),
    "legion_cache_max_id_num": (c_i32, [c_p, c_i32]),
    "legion_cache_topo_transactions": (ctypes.c_uint64, [c_p, c_i32]),
    "legion_cache_array": (c_p, [c_p, c_i32, c_i32]),
    "legion_cache_find_topo": (None, [c_p, c_i32, c_p, c_p, c_i32, c_p, c_p]),
    "legion_cache_find_feat": (None, [c_p, c_i32, c_p, c_p, c_p, c_p, c_i32]),
    # 3. server / ipc
    "NewGPUServer": (c_p, []),
    "legion_server_initialize": (None, [c_p, c_i32, P_I32, c_i32, c_i32]),
    "legion_server_presc": (None, [c_p, c_i32]),
    "legion_server_run": (None, [c_p]),
    "legion_server_finalize": (None, [c_p]),
    "legion_run": (c_i32, [P_I32, c_i32, c_i32, c_i32, c_i32]),
    "NewIPCEnv": (c_p, [c_i32]),
    "legion_ipc_coordinate": (None, [c_p, c_i32, P_I32, P_I32, P_I32, c_i32, c_i32]),
    "legion_ipc_train_step": (c_i32, [c_p]),
    "legion_ipc_max_step": (c_i32, [c_p]),
    "legion_ipc_current_mode": (c_i32, [c_p, c_i32]),
    "legion_ipc_local_batch_id": (c_i32, [c_p, c_i32]),
    "legion_ipc_current_batchsize": (c_i32, [c_p, c_i32, c_i32]),
    "legion_ipc_finalize": (None, [c_p]),
    "legion_group_create": (c_p, [ctypes.POINTER(c_p), c_i32]),
    "legion_group_set_iter_state": (None, [c_p, c_p]),
    "legion_group_destroy": (None, [c_p]),
    "legion_enqueue_group": (None, [c_p, c_p, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_i32, P_I32, c_i32]),
    "legion_pipeline_create": (c_p, [c_p, c_p, c_p, c_i32, c_i32, P_I32, c_i32, c_i32, c_i32, c_i64, c_i32]),
    "legion_pipeline_submit": (c_i32, [c_p, c_i32, c_i32]),
    "legion_pipeline_submit_n": (c_i32, [c_p, c_i32, c_i32, c_i32]),
    "legion_enqueue_group_n": (None, [c_p, c_p, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_i32, P_I32, c_i32]),
    "legion_pipeline_wait": (None, [c_p, c_i32]),
    "legion_pipeline_pool": (c_p, [c_p, c_i32, c_i32]),
    "legion_pipeline_destroy": (None, [c_p]),
    "legion_pipeline_bulk_enable": (c_i32, [c_p]),
    "legion_pipeline_bulk_export": (c_i32, [c_p, c_p, c_i32]),
    "legion_pipeline_bulk_import": (c_i32, [c_p, c_p]),
    "legion_pipeline_bulk_link": (c_i32, [c_p, c_p]),
    "legion_pipeline_bulk_phase_a": (c_i32, [c_p, c_i32, c_i32, c_i32, c_i32]),
    "legion_pipeline_bulk_phase_b": (None, [c_p, c_i32]),
    "legion_pipeline_bulk_listed": (c_i64, [c_p, c_i32]),
    "legion_pipeline_profile_begin": (None, [c_p]),
    "legion_pipeline_profile_end": (None, [c_p]),
    "legion_pipeline_profile_read": (c_i32, [c_p, P_I32, ctypes.POINTER(ctypes.c_double),
                                             ctypes.POINTER(ctypes.c_int64), c_i32]),
    "legion_pipeline_regather_last": (c_i32, [c_p, c_i32, c_i32, c_i32, ctypes.POINTER(ctypes.c_double)]),
    # 4. kernel-level
    "legion_gather_rows": (None, [c_p, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_p, c_p, c_p, c_p, c_i32]),
    "legion_draw_batch": (None, [c_p, c_p, c_p, c_p, c_i32]),
    "legion_pool_profile_begin": (None, [c_p, c_i32]),
    "legion_pool_profile_end": (c_i32, [c_p, ctypes.POINTER(ctypes.c_float), P_I32, c_i32]),
    # 5. synthetic workloads
    "legion_synth_rmat_edges": (None, [c_p, c_i32, c_i64, c_u64, c_p, c_p]),
    "legion_synth_rmat_edges_scrambled": (None, [c_p, c_i32, c_i64, c_u64, c_p, c_p, c_u64]),
    "legion_synth_features": (None, [c_p, c_p, c_i64, c_i64, c_i32, c_u64]),
    "legion_synth_feature_check": (None, [c_p, c_p, c_p, c_i64, c_i32, c_u64, c_p]),
    "legion_consume_batch": (None, [c_p, c_p, c_i64, c_p, c_p, c_i64, c_p]),
    "legion_link_counters": (c_i32, [c_i32, P_U64, P_U64]),
    "legion_link_counters_ex": (c_i32, [c_i32, c_p]),
    "legion_link_counters_from": (c_i32, [c_i32, c_i32, c_p]),
    "legion_cache_peer_transactions": (ctypes.c_uint64, [c_p, c_i32]),
    "legion_cache_gather_stats3": (None, [c_p, c_i32, P_U64]),
    "legion_cache_gather_stats_enable": (None, [c_p, c_i32]),
    # 6. tuning
    "legion_tuning_from_env": (None, []),
    "legion_tuning_get": (None, [c_p]),
    "legion_tuning_set": (None, [c_p]),
    "legion_host_alloc": (c_p, [c_i64, ctypes.POINTER(c_p)]),
    "legion_host_free": (None, [c_p]),
    "legion_set_device_base": (None, [c_i32]),
    "legion_get_device_base": (c_i32, []),
    "legion_version": (ctypes.c_char_p, []),
    "legion_device_count": (c_i32, []),
}



class LinkCounters(ctypes.Structure):          # LegionLinkCounters
    _fields_ = [("pcie_bytes", c_u64), ("xgmi_read_bytes", c_u64), ("xgmi_write_bytes", c_u64),
                ("xgmi_read_bytes_link", c_u64 * 8), ("xgmi_write_bytes_link", c_u64 * 8),
                ("format_revision", c_i32), ("content_revision", c_i32), ("pci_bus_id", ctypes.c_char * 32),
                ("source", c_i32), ("reserved", c_i32)]


class Tuning(ctypes.Structure):                # LegionTuning (include/legion_hip.h section 6)
    _fields_ = [(n, c_i32) for n in (
        "lds_small_buckets", "lds_part_wg", "sample_max_wg", "lds_known_cap", "lds_claim_cap", "col_slots", "gather_rows_per_wg", "peer_gather",
        "arena_scatter_mb", "weave_priority", "markers", "runner_graph", "runner_lanes", "runner_slots", "runner_handover", "runner_ho_stream",
        "runner_spin_us", "runner_overflow", "runner_stats", "shm_mirror", "table_placement", "hotness_reduce", "link_counters")] + \
        [("link_counter_values", c_u64 * 2)]


_lib = None


def load():
    """Returns the loaded library with argtypes set; raises ImportError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m legion_amd.build` "
            "(hipcc --offload-arch=gfx950).  legion_amd has no CPU fallback.")
    try:
        lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    except OSError as e:  # pragma: no cover
        raise ImportError(f"cannot load {LIB_PATH}: {e}.  legion_amd has no CPU fallback.") from e
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError here = header/library mismatch: fail loudly
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib
