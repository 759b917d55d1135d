"""Host-side Python mirror of the reference's operator interface for the hot path.

Same names and argument meaning as the reference (SS = sampling_server/src):
  BatchGenerate / RandomSample / FeatureCacheLookup / IOSubmit / IOComplete
                                         SS/engine/operator_impl.cuh:11-63
  GraphStorage / FeatureStorage          SS/storage/graph_storage.cuh:7-24, feature_storage.cuh:6-34
  MemoryPool                             SS/engine/memorypool.cuh:20-221
  UnifiedCache                           SS/cache/cache.cuh:66-177
Everything computes inside liblegion_hip.so (HIP kernels for gfx950) through the C ABI in
include/legion_hip.h; PyTorch is used only to own device memory, streams and torch.distributed.
There is no CPU path: constructing any of these without a GPU-backed library raises.
"""
import ctypes

import numpy as np
import torch

from . import lib as _libmod

INTERBATCH_CON = 2
INTRABATCH_CON = 3
TRAINMODE, VALIDMODE, TESTMODE = 0, 1, 2
CACHEMISS_FLAG = -2

_TYPESTR = {torch.int32: "<i4", torch.int64: "<i8", torch.float32: "<f4", torch.int8: "|i1",
            torch.uint8: "|u1"}


class _RawDevice:
    def __init__(self, ptr, shape, dtype):
        self.__cuda_array_interface__ = {
            "shape": tuple(int(s) for s in shape), "typestr": _TYPESTR[dtype],
            "data": (int(ptr), False), "version": 2, "strides": None}


def device_view(ptr, shape, dtype, device):
    """Zero-copy torch view of raw device memory owned by the library."""
    n = 1
    for s in shape:
        n *= int(s)
    if not ptr or n == 0:
        return torch.empty(tuple(shape), dtype=dtype, device=device)
    return torch.as_tensor(_RawDevice(ptr, shape, dtype), device=device)


def set_device_base(base):
    """One process per GPU: this process's logical GPU 0 is physical GPU `base` (LOCAL_RANK)."""
    _libmod.load().legion_set_device_base(int(base))


def set_local_device(dev):
    """A clique spread over processes: this process owns logical GPU `dev` only (dev = its rank)."""
    _libmod.load().legion_set_local_device(int(dev))


def link_counters(dev_id=0):
    """(PCIe bytes, xGMI bytes) moved by logical GPU dev_id since boot, from the driver's gpu_metrics table, or None
    when the table is not readable / has an unknown revision (legion_hip.h: legion_link_counters)."""
    a, b = ctypes.c_uint64(0), ctypes.c_uint64(0)
    ok = _libmod.load().legion_link_counters(int(dev_id), ctypes.byref(a), ctypes.byref(b))
    return (int(a.value), int(b.value)) if ok else None


def link_counters_ex(dev_id=0, source=0):
    """The whole table as a dict (legion_hip.h: LegionLinkCounters) plus 'supported': PCIe bytes, xGMI bytes read / written
    in total and per link, the gpu_metrics revision found, the GPU's PCI bus id and which decoder produced the numbers
    (source: 0 = rocm_smi_lib's versioned decoder first, the byte-offset parser second; 1 / 2 = that one only)."""
    c = _libmod.LinkCounters()
    ok = _libmod.load().legion_link_counters_from(int(dev_id), int(source), ctypes.byref(c))
    return {"supported": bool(ok), "source": {0: None, 1: "rocm_smi_lib", 2: "sysfs gpu_metrics by offset"}.get(int(c.source)), "pcie_bytes": int(c.pcie_bytes), "xgmi_read_bytes": int(c.xgmi_read_bytes),
            "xgmi_write_bytes": int(c.xgmi_write_bytes), "xgmi_read_bytes_link": [int(x) for x in c.xgmi_read_bytes_link],
            "xgmi_write_bytes_link": [int(x) for x in c.xgmi_write_bytes_link],
            "gpu_metrics_revision": f"{int(c.format_revision)}.{int(c.content_revision)}",
            "pci_bus_id": c.pci_bus_id.decode(errors="replace")}


def tuning():
    """The library's current LegionTuning as a dict."""
    t = _libmod.Tuning()
    _libmod.load().legion_tuning_get(ctypes.byref(t))
    return {n: (list(getattr(t, n)) if n == "link_counter_values" else int(getattr(t, n))) for n, _ in t._fields_}


def set_tuning(**fields):
    """Installs programmatic tuning values (kept until tuning_from_env() is called): set_tuning(runner_lanes=4, ...)."""
    t = _libmod.Tuning()
    L = _libmod.load()
    L.legion_tuning_get(ctypes.byref(t))
    for k, v in fields.items():
        if k == "link_counter_values":
            t.link_counter_values[0], t.link_counter_values[1] = int(v[0]), int(v[1])
        else:
            if not hasattr(t, k):
                raise KeyError(k)
            setattr(t, k, int(v))
    L.legion_tuning_set(ctypes.byref(t))


def tuning_from_env():
    _libmod.load().legion_tuning_from_env()


def _torch_device(dev_id):
    base = int(_libmod.load().legion_get_device_base())
    return torch.device("cuda", (base + int(dev_id)) % max(torch.cuda.device_count(), 1))


class PinnedArray:
    """A numpy-visible array in mapped pinned host memory that the GPU reads in place over PCIe: the
    spill-over tier for tables that do not fit HBM (the reference's only tier for the full CSR/features)."""

    def __init__(self, array):
        array = np.ascontiguousarray(array)
        self._lib = _libmod.load()
        host = ctypes.c_void_p()
        self.dev_ptr = self._lib.legion_host_alloc(int(array.nbytes), ctypes.byref(host))
        self.host_ptr = host.value
        self.shape, self.dtype = array.shape, array.dtype
        ctypes.memmove(self.host_ptr, array.ctypes.data, array.nbytes)

    @classmethod
    def empty(cls, shape, dtype):
        """Uninitialised mapped pinned array (fill it through .tensor(device) or .numpy())."""
        self = cls.__new__(cls)
        self._lib = _libmod.load()
        self.shape, self.dtype = tuple(int(s) for s in shape), np.dtype(dtype)
        nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        host = ctypes.c_void_p()
        self.dev_ptr = self._lib.legion_host_alloc(nbytes, ctypes.byref(host))
        self.host_ptr = host.value
        return self

    def numpy(self):
        n = int(np.prod(self.shape, dtype=np.int64))
        buf = (ctypes.c_char * (n * self.dtype.itemsize)).from_address(self.host_ptr)
        return np.frombuffer(buf, dtype=self.dtype, count=n).reshape(self.shape)

    def tensor(self, device):
        tdt = {np.dtype(np.int32): torch.int32, np.dtype(np.int64): torch.int64, np.dtype(np.float32): torch.float32}[self.dtype]
        return device_view(self.dev_ptr, self.shape, tdt, device)

    def close(self):
        if self.host_ptr:
            self._lib.legion_host_free(ctypes.c_void_p(self.host_ptr))
            self.host_ptr = None


def _stream_handle(stream=None):
    s = stream if stream is not None else torch.cuda.current_stream()
    return ctypes.c_void_p(s.cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _i32_array(values):
    arr = (ctypes.c_int32 * len(values))(*[int(v) for v in values])
    return arr


class GraphStorage:
    """Full CSR (slot P of the pointer tables) from device tensors: indptr int64[N+1], col int32[E]."""

    def __init__(self, partition_count, indptr, col):
        assert indptr.dtype == torch.int64 and col.dtype == torch.int32
        assert indptr.is_cuda and col.is_cuda and indptr.is_contiguous() and col.is_contiguous()
        self._lib = _libmod.load()
        self.indptr, self.col = indptr, col          # keep alive
        self.partition_count = int(partition_count)
        self.node_num = int(indptr.numel() - 1)
        self.edge_num = int(col.numel())
        self.handle = self._lib.legion_graph_create(self.partition_count, self.node_num, self.edge_num,
                                                    _ptr(indptr), _ptr(col))

    def column_slots(self, dev_id=0):
        """True when logical GPU dev_id samples from the {neighbour id, feature-cache slot} copy of the column array."""
        return bool(self._lib.legion_graph_column_slots(self.handle, int(dev_id)))

    def cached_csr(self, dev_id, capacity):
        """(index int64[capacity + 1], dst int32[index[capacity]]) of the CSR GPU dev_id caches after a fill, as device views."""
        a, b = ctypes.c_void_p(), ctypes.c_void_p()
        self._lib.legion_graph_cached_csr(self.handle, int(dev_id), ctypes.byref(a), ctypes.byref(b))
        dev = _torch_device(dev_id)
        index = device_view(a.value, (int(capacity) + 1,), torch.int64, dev)
        n = int(index[-1].item()) if capacity >= 0 and a.value else 0
        return index, device_view(b.value, (n,), torch.int32, dev)

    def close(self):
        if self.handle:
            self._lib.legion_graph_destroy(self.handle)
            self.handle = None


class FeatureStorage:
    """Full feature table (device tensor float32[N, D], HBM or mapped pinned) + per-GPU seed sets."""

    def __init__(self, partition_count, features, total_num_nodes=None, float_feature_len=None):
        self._lib = _libmod.load()
        self.features = features
        if features is not None:
            assert features.dtype == torch.float32 and features.is_contiguous()
            total_num_nodes = features.shape[0] if total_num_nodes is None else total_num_nodes
            float_feature_len = features.shape[1] if float_feature_len is None else float_feature_len
        self.total_num_nodes = int(total_num_nodes)
        self.float_feature_len = int(float_feature_len)
        self.partition_count = int(partition_count)
        self.handle = self._lib.legion_feature_create(self.partition_count, self.total_num_nodes,
                                                      self.float_feature_len, _ptr(features))
        self.set_sizes = {}

    def set_ids(self, dev_id, mode, ids, labels=None):
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        lab = None if labels is None else np.ascontiguousarray(labels, dtype=np.int32)
        self._lib.legion_feature_set_ids(self.handle, int(dev_id), int(mode),
                                         ids.ctypes.data_as(ctypes.c_void_p),
                                         lab.ctypes.data_as(ctypes.c_void_p) if lab is not None else None,
                                         int(ids.size))
        self.set_sizes[(int(dev_id), int(mode))] = int(ids.size)

    def close(self):
        if self.handle:
            self._lib.legion_feature_destroy(self.handle)
            self.handle = None


class MemoryPool:
    _BUF = {"sampled_ids": (0, torch.int32), "float_features": (1, torch.float32),
            "labels": (2, torch.int32), "agg_src_off": (3, torch.int32), "agg_dst_off": (4, torch.int32),
            "node_counter": (5, torch.int32), "edge_counter": (6, torch.int32),
            "agg_src_ids": (7, torch.int32), "agg_dst_ids": (8, torch.int32),
            "cache_search_buffer": (9, torch.int32), "tmp_part_ind": (10, torch.int8),
            "tmp_part_off": (11, torch.int32), "position_map": (12, torch.int32), "node_slot": (13, torch.int32)}

    def __init__(self, dev_id, total_num_nodes, batch_size, fanout, float_feature_len, pipeline_depth=1):
        self._lib = _libmod.load()
        self.dev_id = int(dev_id)
        self.device = _torch_device(self.dev_id)
        self.total_num_nodes = int(total_num_nodes)
        self.batch_size = int(batch_size)
        self.fanout = [int(f) for f in fanout]
        self.float_feature_len = int(float_feature_len)
        self.feature_rows = 0
        self.handle = self._lib.legion_pool_create(self.dev_id, self.total_num_nodes, self.batch_size,
                                                   _i32_array(self.fanout), len(self.fanout),
                                                   self.float_feature_len, int(pipeline_depth))
        self.num_ids = int(self._lib.legion_pool_num_ids(self.handle))

    @classmethod
    def _borrowed(cls, handle, dev_id, total_num_nodes, batch_size, fanout, float_feature_len, feature_rows):
        """View of a pool owned by a Pipeline lane (close() is a no-op)."""
        self = cls.__new__(cls)
        self._lib = _libmod.load()
        self.dev_id = int(dev_id)
        self.device = _torch_device(self.dev_id)
        self.total_num_nodes, self.batch_size = int(total_num_nodes), int(batch_size)
        self.fanout = [int(f) for f in fanout]
        self.float_feature_len, self.feature_rows = int(float_feature_len), int(feature_rows)
        self.handle = handle
        self.num_ids = int(self._lib.legion_pool_num_ids(handle))
        self._borrowed_handle = True
        return self

    def alloc_features(self, rows):
        self.feature_rows = int(rows)
        self._lib.legion_pool_alloc_features(self.handle, self.feature_rows)

    def set_current_pipe(self, pipe):
        self._lib.legion_pool_set_current_pipe(self.handle, int(pipe))

    def buffer(self, name):
        which, dtype = self._BUF[name]
        ptr = self._lib.legion_pool_buffer(self.handle, which)
        if name in ("node_counter", "edge_counter"):
            shape = (16,)
        elif name == "labels":
            shape = (self.batch_size,)
        elif name == "float_features":
            shape = (self.feature_rows, self.float_feature_len)
        elif name == "position_map":
            shape = (self.total_num_nodes,)
        else:
            shape = (self.num_ids,)
        return device_view(ptr, shape, dtype, self.device)

    def lds_buckets(self):
        """Hash buckets per lane of the first-touch de-duplication (8, 16, 64 or 256 by the pool's largest hop)."""
        return int(self._lib.legion_pool_lds_buckets(self.handle))

    def state_bytes(self):
        return int(self._lib.legion_pool_state_bytes(self.handle))

    def error(self):
        """Sticky LG_ERR_* bits raised on the device for this pool (0 = none)."""
        return int(self._lib.legion_pool_error(self.handle))

    def profile_begin(self, max_ops):
        self._lib.legion_pool_profile_begin(self.handle, int(max_ops))

    def profile_end(self, max_ops):
        """(ms, op_id) arrays of the gathers timed since profile_begin; synchronise the stream first."""
        ms = (ctypes.c_float * max_ops)()
        ops = (ctypes.c_int32 * max_ops)()
        n = self._lib.legion_pool_profile_end(self.handle, ms, ops, int(max_ops))
        return np.array(ms[:n], dtype=np.float64), np.array(ops[:n], dtype=np.int32)

    def close(self):
        if self.handle and not getattr(self, "_borrowed_handle", False):
            self._lib.legion_pool_destroy(self.handle)
        self.handle = None


class LaneGroup:
    """G pools served by every launch (grid.y = G); lane i produces batch counter0 + i."""

    def __init__(self, pools):
        self._lib = _libmod.load()
        self.pools = list(pools)
        arr = (ctypes.c_void_p * len(self.pools))(*[p.handle for p in self.pools])
        self.handle = self._lib.legion_group_create(arr, len(self.pools))

    def enqueue(self, strm_hdl, graph, feature, cache, batch_size, counter0, dev_id, mode, fanout):
        self._lib.legion_enqueue_group(_stream_handle(strm_hdl), graph.handle, feature.handle,
                                       cache.handle if cache else None, self.handle, int(batch_size), int(counter0),
                                       int(dev_id), int(mode), _i32_array(fanout), len(fanout))

    def close(self):
        if self.handle:
            self._lib.legion_group_destroy(self.handle)
            self.handle = None


def collective_unique_id():
    """128 bytes that rank 0 creates and every rank needs for collective_init_rank (carry them over any channel)."""
    buf = ctypes.create_string_buffer(128)
    if not _libmod.load().legion_collective_unique_id(buf):
        raise RuntimeError("ncclGetUniqueId failed")
    return buf.raw


def collective_init_rank(unique_id, world, rank, dev_id=0):
    """Joins the library's own RCCL communicator (the hotness all-reduce of a one-process-per-GPU deployment)."""
    return bool(_libmod.load().legion_collective_init_rank(ctypes.create_string_buffer(unique_id, 128), int(world), int(rank), int(dev_id)))


def collective_destroy():
    _libmod.load().legion_collective_destroy()


class Pipeline:
    """`slots` groups of `group_size` mini-batches in flight on one GPU, each group replayed as one
    hipGraph (pipeline.hip).  submit(counter0) enqueues batches counter0 .. counter0+group_size-1.
    arena: False = every lane's arrays are allocations of their own; True = all lanes' trainer-visible arrays in ONE arena built from
    shuffled physical chunks (LegionTuning.arena_scatter_mb: the gathers' rows land all over the HBM); "shared" = an arena that other GPUs and
    processes can reach (peer_gather = bulk: owners push rows into it) -- the same shuffled chunks, created exportable; other GPUs of this
    process are granted access at bulk_link, other processes map them from file descriptors at bulk_import ("plain": the old name)."""

    def __init__(self, graph, feature, cache, dev_id, batch_size, fanout, group_size, feature_rows, use_graph=True,
                 slots=2, overlap=False, split=False, weave=False, arena=False):      # (split: accepted and ignored -- removed in round 5)
        self._lib = _libmod.load()
        self.group_size, self.slots = int(group_size), int(slots)
        self.fanout = [int(f) for f in fanout]
        self.handle = self._lib.legion_pipeline_create(graph.handle, feature.handle, cache.handle, int(dev_id),
                                                       int(batch_size), _i32_array(self.fanout), len(self.fanout),
                                                       self.group_size, self.slots, int(feature_rows),
                                                       (1 if use_graph else 0) | (2 if overlap else 0) | (4 if split else 0) |
                                                       (16 if weave else 0) | (32 if arena else 0) | (64 if arena in ("shared", "plain") else 0))
        self.pools = [[MemoryPool._borrowed(self._lib.legion_pipeline_pool(self.handle, s, g), dev_id,
                                            feature.total_num_nodes, batch_size, fanout, feature.float_feature_len,
                                            feature_rows) for g in range(self.group_size)]
                      for s in range(self.slots)]

    def submit(self, counter0, mode=TRAINMODE, n_active=None):
        if n_active is None or n_active >= self.group_size:
            return int(self._lib.legion_pipeline_submit(self.handle, int(counter0), int(mode)))
        return int(self._lib.legion_pipeline_submit_n(self.handle, int(counter0), int(mode), int(n_active)))

    def run_range(self, first, count, mode=TRAINMODE, wrap=None):
        """Submits batches first .. first+count-1 as full groups plus, if needed, one partial group.
        `wrap` (a number of batches): batch indices are taken modulo it -- another epoch over the same seed set, as
        the reference's schedule does with GetLocalBatchId (ipc_service.cu:213-228); a group never straddles the wrap.
        Returns (slot, first batch, lanes) of the last group submitted."""
        k, last = 0, None
        while k < count:
            b = (first + k) % wrap if wrap else first + k
            n = min(self.group_size, count - k, (wrap - b) if wrap else count)
            last = (self.submit(b, mode, n), b, n)
            k += n
        return last

    def wait(self, slot=-1):
        self._lib.legion_pipeline_wait(self.handle, int(slot))

    # ---- peer_gather = bulk (pipeline.hip): owner-bucketed transfer of the rows a striped gather needs from other members ----
    def bulk_enable(self):
        """Needs arena="shared".  Allocates this GPU's per-owner request lists (one set per slot)."""
        if not self._lib.legion_pipeline_bulk_enable(self.handle):
            raise RuntimeError("legion_pipeline_bulk_enable failed (pipeline not created with arena='plain'?)")

    def bulk_export(self):
        """Bytes the other members need (IPC handles of the lane arena and the lists)."""
        buf = ctypes.create_string_buffer(512)
        n = int(self._lib.legion_pipeline_bulk_export(self.handle, buf, 512))
        if n <= 0:
            raise RuntimeError("legion_pipeline_bulk_export failed")
        return buf.raw[:n]

    def bulk_import(self, handles):
        """A member in ANOTHER process."""
        if not self._lib.legion_pipeline_bulk_import(self.handle, ctypes.create_string_buffer(handles, len(handles))):
            raise RuntimeError("legion_pipeline_bulk_import failed")

    def bulk_link(self, other):
        """A member in THIS process (logical GPUs of a test, a thread per GPU)."""
        if not self._lib.legion_pipeline_bulk_link(self.handle, other.handle):
            raise RuntimeError("legion_pipeline_bulk_link failed")

    def bulk_phase_a(self, counter0, mode=TRAINMODE, n_active=None, batch_size=0):
        """Sampler + bucket pass + gather of everything that is not another member's stripe; returns the slot.  The caller
        barriers over the clique, calls bulk_phase_b(slot) on every member, and barriers again."""
        return int(self._lib.legion_pipeline_bulk_phase_a(self.handle, int(counter0), int(mode),
                                                          int(n_active) if n_active else self.group_size, int(batch_size)))

    def bulk_phase_b(self, slot):
        self._lib.legion_pipeline_bulk_phase_b(self.handle, int(slot))

    def bulk_listed(self, slot):
        """Rows this GPU listed for other members in `slot` (what phase B sends towards it)."""
        return int(self._lib.legion_pipeline_bulk_listed(self.handle, int(slot)))

    def profile_begin(self):
        self._lib.legion_pipeline_profile_begin(self.handle)

    def profile_end(self):
        self._lib.legion_pipeline_profile_end(self.handle)

    def regather_last(self, slot, repeats=5, n_active=0):
        """The last op's gather of the group sitting in `slot`, `repeats` more times over the lanes as they stand: ms per launch
        (HIP events on the slot's stream).  legion_hip.h: legion_pipeline_regather_last."""
        ms = (ctypes.c_double * int(repeats))()
        n = self._lib.legion_pipeline_regather_last(self.handle, int(slot), int(n_active), int(repeats), ms)
        return [float(ms[i]) for i in range(n)]

    def profile_read(self):
        """{gather op id: (summed ms, launches)} for every batch waited for since profile_begin()."""
        ops = (ctypes.c_int32 * 16)()
        ms = (ctypes.c_double * 16)()
        cnt = (ctypes.c_int64 * 16)()
        n = self._lib.legion_pipeline_profile_read(self.handle, ops, ms, cnt, 16)
        return {int(ops[i]): (float(ms[i]), int(cnt[i])) for i in range(n)}

    def close(self):
        if self.handle:
            self._lib.legion_pipeline_destroy(self.handle)
            self.handle = None


class UnifiedCache:
    _ARR = {"QF": (0, torch.int32), "QT": (1, torch.int32), "AF": (2, torch.int64), "AT": (3, torch.int64),
            "node_access_time": (4, torch.int64), "edge_access_time": (5, torch.int64),
            "node_map": (6, torch.int32), "edge_index_map": (7, torch.int8), "edge_offset_map": (8, torch.int32)}

    def __init__(self, cache_memory, float_feature_len, train_step, device_count, total_num_nodes):
        self._lib = _libmod.load()
        self.device_count = int(device_count)
        self.total_num_nodes = int(total_num_nodes)
        self.handle = self._lib.legion_cache_create(int(cache_memory), int(float_feature_len), int(train_step),
                                                    self.device_count, self.total_num_nodes)

    def init_controller(self, dev_id):
        self._lib.legion_cache_init_controller(self.handle, int(dev_id))

    def candidate_selection(self, cache_agg_mode, graph, world_reduced=False):
        self._lib.legion_cache_candidate_selection(self.handle, int(cache_agg_mode), graph.handle,
                                                   1 if world_reduced else 0)

    def allreduce_hotness(self, dev_id=0):
        """One process per GPU: RCCL all-reduce (issued by the library, collective_init_rank's communicator) of this GPU's two
        access-counter arrays in place.  Returns (world size the collective ran over -- 0 on failure --, milliseconds)."""
        ms = ctypes.c_double(0)
        world = int(self._lib.legion_cache_allreduce_hotness(self.handle, int(dev_id), ctypes.byref(ms)))
        return world, float(ms.value)

    def hotness_reduce_path(self, dev_id=0):
        """How the last candidate_selection summed the clique's counters: 'none', 'p2p' (leader loop) or 'rccl'."""
        return ("none", "p2p", "rccl")[int(self._lib.legion_cache_hotness_reduce_path(self.handle, int(dev_id)))]

    def cost_model(self, feature, graph, counters=(0, 0), train_step=0):
        cnt = (ctypes.c_uint64 * 2)(int(counters[0]), int(counters[1]))
        self._lib.legion_cache_cost_model(self.handle, feature.handle, graph.handle, cnt, int(train_step))

    def set_replica_memory(self, nbytes):
        """Every member of a striped clique also keeps the clique's hottest rows locally, as many as nbytes hold."""
        self._lib.legion_cache_set_replica_memory(self.handle, int(nbytes))

    def replica_rows(self, dev_id=0):
        return int(self._lib.legion_cache_replica_rows(self.handle, int(dev_id)))

    def gather_stats(self, dev_id=0):
        """(rows read through a stripe pointer, rows read from the local replica) so far; the first call enables counting."""
        out = (ctypes.c_uint64 * 2)()
        self._lib.legion_cache_gather_stats(self.handle, int(dev_id), out)
        return int(out[0]), int(out[1])

    def gather_stats3(self, dev_id=0):
        """(rows through a stripe pointer, rows from the local replica, rows from a PEER's stripe) so far."""
        out = (ctypes.c_uint64 * 3)()
        self._lib.legion_cache_gather_stats3(self.handle, int(dev_id), out)
        return int(out[0]), int(out[1]), int(out[2])

    def gather_stats_enable(self, on):
        self._lib.legion_cache_gather_stats_enable(self.handle, 1 if on else 0)

    def peer_transactions(self, dev_id=0):
        """64-byte transactions read from other members' stripes so far (the computed stand-in for the xGMI counter)."""
        return int(self._lib.legion_cache_peer_transactions(self.handle, int(dev_id)))

    def set_capacity(self, node_capacity, edge_capacity):
        self._lib.legion_cache_set_capacity(self.handle, int(node_capacity), int(edge_capacity))

    def fill_up(self, feature, graph):
        self._lib.legion_cache_fill_up(self.handle, feature.handle, graph.handle)

    def hybrid_init(self, feature, graph, cpu_cache_capacity, gpu_cache_capacity, miss_from_table=True):
        """The hybrid CPU-cache / GPU-cache tier instead of candidate_selection + cost_model + fill_up
        (UnifiedCache::HybridInit, SS/cache/cache.cu:614-670)."""
        self._hybrid = (int(cpu_cache_capacity), int(gpu_cache_capacity), int(feature.float_feature_len))
        self._lib.legion_cache_hybrid_init(self.handle, feature.handle, graph.handle, int(cpu_cache_capacity),
                                           int(gpu_cache_capacity), 1 if miss_from_table else 0)

    def hybrid_caches(self, dev_id=0):
        """(CPU cache, GPU cache) of dev_id after hybrid_init as float32 tensors [capacity, D] (device views)."""
        cpu_cap, gpu_cap, dim = self._hybrid
        dev = _torch_device(dev_id)
        cpu = device_view(self._lib.legion_cache_hybrid_cpu_cache(self.handle, int(dev_id)), (cpu_cap, dim), torch.float32, dev)
        gpu = device_view(self._lib.legion_cache_feature_cache(self.handle, int(dev_id)), (gpu_cap, dim), torch.float32, dev)
        return cpu, gpu

    def fill_up_distributed(self, feature, graph, rank, world, max_ids_all, all_gather_bytes):
        """FillUp of a clique spread over `world` processes (this process owns member `rank`):
        local stripe -> export 3 IPC handles -> all-gather -> open the peers' -> link.
        `all_gather_bytes(b: bytes) -> list[bytes]` is the caller's collective (torch.distributed)."""
        self._lib.legion_cache_fill_up_local(self.handle, feature.handle, graph.handle)
        mine = ctypes.create_string_buffer(192)
        self._lib.legion_cache_export(self.handle, graph.handle, int(rank), mine)
        everyone = all_gather_bytes(mine.raw)
        self._peer_handles = [ctypes.create_string_buffer(b, 192) for b in everyone]
        for peer in range(world):
            if peer != rank:
                self._lib.legion_cache_import_peer(self.handle, graph.handle, int(rank), peer, self._peer_handles[peer])
        self._lib.legion_cache_fill_up_link(self.handle, feature.handle, graph.handle)

    def set_peer_max_ids(self, max_ids_all):
        arr = _i32_array(max_ids_all)
        self._lib.legion_cache_set_peer_max_ids(self.handle, arr, len(max_ids_all))

    def node_capacity(self, dev_id=0):
        return int(self._lib.legion_cache_node_capacity(self.handle, int(dev_id)))

    def edge_capacity(self, dev_id=0):
        return int(self._lib.legion_cache_edge_capacity(self.handle, int(dev_id)))

    def max_id_num(self, dev_id=0):
        return int(self._lib.legion_cache_max_id_num(self.handle, int(dev_id)))

    def topo_transactions(self, dev_id):
        """64-byte transactions of GPU dev_id's PreSC topology reads (the PCM counter of the paper; see legion_hip.h)."""
        return int(self._lib.legion_cache_topo_transactions(self.handle, int(dev_id)))

    def find_topo(self, dev_id, input_ids):
        """(partition_index int8, partition_offset int32) for a device int32 tensor of vertex ids."""
        n = int(input_ids.numel())
        ind = torch.empty(n, dtype=torch.int8, device=input_ids.device)
        off = torch.empty(n, dtype=torch.int32, device=input_ids.device)
        self._lib.legion_cache_find_topo(self.handle, int(dev_id), _stream_handle(None), _ptr(input_ids), n,
                                         _ptr(ind), _ptr(off))
        return ind, off

    def array(self, name, dev_id=0):
        which, dtype = self._ARR[name]
        ptr = self._lib.legion_cache_array(self.handle, int(dev_id), which)
        return device_view(ptr, (self.total_num_nodes,), dtype, _torch_device(dev_id))

    def close(self):
        if self.handle:
            self._lib.legion_cache_destroy(self.handle)
            self.handle = None


# ---- the five operators, reference names and argument order ------------------------------------
def BatchGenerate(strm_hdl, feature, cache, memorypool, batch_size, counter, part_id, dev_id, mode,
                  is_presc, hop_num):
    _libmod.load().BatchGenerate(_stream_handle(strm_hdl), feature.handle, cache.handle if cache else None,
                                 memorypool.handle, int(batch_size), int(counter), int(part_id), int(dev_id),
                                 int(mode), bool(is_presc), int(hop_num))


def RandomSample(strm_hdl, graph, cache, memorypool, count, dev_id, op_id, is_presc):
    _libmod.load().RandomSample(_stream_handle(strm_hdl), graph.handle, cache.handle if cache else None,
                                memorypool.handle, int(count), int(dev_id), int(op_id), bool(is_presc))


def FeatureCacheLookup(strm_hdl, cache, memorypool, op_id, dev_id):
    _libmod.load().FeatureCacheLookup(_stream_handle(strm_hdl), cache.handle, memorypool.handle, int(op_id),
                                      int(dev_id))


def IOSubmit(strm_hdl, feature, memorypool, op_id, dev_id):
    _libmod.load().IOSubmit(_stream_handle(strm_hdl), feature.handle, memorypool.handle, int(op_id), int(dev_id))


def IOComplete(strm_hdl, cache, memorypool, dev_id, mode):
    _libmod.load().IOComplete(_stream_handle(strm_hdl), cache.handle if cache else None, memorypool.handle,
                              int(dev_id), int(mode))


def enqueue_batch(strm_hdl, graph, feature, cache, memorypool, batch_size, counter, dev_id, mode, is_presc,
                  fanout):
    """All ops of one mini-batch in GPURunner::RunOnce order (SS/engine/server.cu:302-332)."""
    _libmod.load().legion_enqueue_batch(_stream_handle(strm_hdl), graph.handle, feature.handle,
                                        cache.handle if cache else None, memorypool.handle, int(batch_size),
                                        int(counter), int(dev_id), int(mode), bool(is_presc),
                                        _i32_array(fanout), len(fanout))


def read_batch(memorypool):
    """Host copy of everything a trainer (and the parity tests) can observe about the current batch."""
    nc = memorypool.buffer("node_counter").cpu().numpy().copy()
    ec = memorypool.buffer("edge_counter").cpu().numpy().copy()
    hop_num = int(nc[INTRABATCH_CON * 3 - 1])
    n_nodes = int(nc[INTRABATCH_CON * 3 + hop_num])
    n_edges = int(ec[INTRABATCH_CON * 3 + hop_num])
    out = {"node_counter": nc, "edge_counter": ec, "hop_num": hop_num,
           "sampled_ids": memorypool.buffer("sampled_ids")[:max(n_nodes, 0)].cpu().numpy().copy(),
           "labels": memorypool.buffer("labels")[:max(int(nc[INTRABATCH_CON * 3]), 0)].cpu().numpy().copy(),
           "agg_src_off": memorypool.buffer("agg_src_off")[:n_edges].cpu().numpy().copy(),
           "agg_dst_off": memorypool.buffer("agg_dst_off")[:n_edges].cpu().numpy().copy(),
           "agg_src_ids": memorypool.buffer("agg_src_ids")[:n_edges].cpu().numpy().copy(),
           "agg_dst_ids": memorypool.buffer("agg_dst_ids")[:n_edges].cpu().numpy().copy()}
    if memorypool.feature_rows > 0:
        out["float_features"] = memorypool.buffer("float_features")[:n_nodes].cpu().numpy().copy()
    return out
