/*
 * legion_hip.h -- C ABI of liblegion_hip.so, the MI355X-native (gfx950) drop-in for the hot path
 * of RC4ML/Legion's sampling server: seed batch -> multi-hop CSR neighbour sampling -> node
 * de-duplication -> feature-cache lookup + gather, plus the one-time hotness -> cache-partition ->
 * fill set-up that feeds it.
 *
 * Every entry point cites the reference interface it replaces (paths relative to the reference
 * repository; SS = sampling_server/src).  Signatures carry plain pointers and sizes only: no
 * torch types, no C++ types.  Object arguments are opaque handles to the library's own
 * GraphStorage / FeatureStorage / UnifiedCache / MemoryPool / IPCEnv objects (the reference passes
 * pointers to its C++ classes of the same names through the same positions).
 *
 * Error convention (SS/engine/operator_impl.cu:16-24,141-148): functions return void; a null
 * object prints a message and returns; any HIP error prints "HIP failure file:line: 'msg'" and
 * exits the process.  There is NO CPU fallback anywhere in this library.
 */
#ifndef LEGION_HIP_H
#define LEGION_HIP_H

#include <stdbool.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* SS/include/system_config.cuh:47-57 -- part of the wire contract */
#define LEGION_INTERBATCH_CON 2
#define LEGION_INTRABATCH_CON 3
#define LEGION_MAX_DEVICE 8
#define LEGION_MEMORY_USAGE 7
#define LEGION_TRAINMODE 0
#define LEGION_VALIDMODE 1
#define LEGION_TESTMODE 2
#define LEGION_CACHEMISS_FLAG (-2)

typedef void* legion_stream_t;            /* hipStream_t in the position of cudaStream_t */
typedef struct LegionGraphStorage   LegionGraphStorage;    /* SS/storage/graph_storage.cuh:7-24  */
typedef struct LegionFeatureStorage LegionFeatureStorage;  /* SS/storage/feature_storage.cuh:6-34 */
typedef struct LegionUnifiedCache   LegionUnifiedCache;    /* SS/cache/cache.cuh:66-177 */
typedef struct LegionMemoryPool     LegionMemoryPool;      /* SS/engine/memorypool.cuh:20-221 */
typedef struct LegionIPCEnv         LegionIPCEnv;          /* SS/engine/ipc_service.h:6-33 */
typedef struct LegionServer         LegionServer;          /* SS/engine/server.h:16-23 */

/* =====================================================================================
 * 1. The five operator entry points -- SS/engine/operator_impl.cuh:11-63, same names, same
 *    argument order and meaning.  All work is enqueued on `strm_hdl`; nothing synchronises
 *    with the host (the reference's blocking 64-byte counter read-backs are gone: frontier and
 *    node counts stay on the device and every kernel reads them there).
 * ===================================================================================== */
void BatchGenerate(legion_stream_t strm_hdl, LegionFeatureStorage* feature, LegionUnifiedCache* cache,
                   LegionMemoryPool* memorypool, int32_t batch_size, int32_t counter, int32_t part_id,
                   int32_t dev_id, int32_t mode, bool is_presc, int32_t hop_num);
void RandomSample(legion_stream_t strm_hdl, LegionGraphStorage* graph, LegionUnifiedCache* cache,
                  LegionMemoryPool* memorypool, int32_t count, int32_t dev_id, int32_t op_id,
                  bool is_presc);
void FeatureCacheLookup(legion_stream_t strm_hdl, LegionUnifiedCache* cache, LegionMemoryPool* memorypool,
                        int32_t op_id, int32_t dev_id);
void IOSubmit(legion_stream_t strm_hdl, LegionFeatureStorage* feature, LegionMemoryPool* memorypool,
              int32_t op_id, int32_t dev_id);
void IOComplete(legion_stream_t strm_hdl, LegionUnifiedCache* cache, LegionMemoryPool* memorypool,
                int32_t dev_id, int32_t mode);

/* One whole mini-batch in the op order of GPURunner::RunOnce / RunPreSc (SS/engine/server.cu:285-332)
 * without the IPC hand-off.  In PreSC mode only ops 0,3,6,...,last run (server.cu:290). */
void legion_enqueue_batch(legion_stream_t strm_hdl, LegionGraphStorage* graph, LegionFeatureStorage* feature,
                          LegionUnifiedCache* cache, LegionMemoryPool* memorypool, int32_t batch_size,
                          int32_t counter, int32_t dev_id, int32_t mode, bool is_presc,
                          const int32_t* fanout, int32_t hop_num);

/* =====================================================================================
 * 2. Object construction from plain buffers (replaces StorageManagement::Initialze's wiring,
 *    SS/storage/storage_management.cu:234-269, for callers that already hold the arrays).
 *    "devptr" = a pointer the GPU can dereference: HBM (hipMalloc / a torch CUDA tensor) or
 *    mapped pinned host memory.  On MI355X the full CSR and, when it fits, the full feature
 *    table live in HBM; the pinned-host tier is the spill-over.
 * ===================================================================================== */

/* GraphStorage: SS/storage/graph_storage.cu:12-73.  Slot `partition_count` of the pointer
 * tables is the full CSR (int64 indptr[N+1], int32 col[E]). */
LegionGraphStorage* legion_graph_create(int32_t partition_count, int32_t node_num, int64_t edge_num,
                                        const int64_t* csr_node_index_devptr,
                                        const int32_t* csr_dst_node_ids_devptr);
void legion_graph_destroy(LegionGraphStorage* g);
/* New in this build ("column slots", LegionTuning.col_slots): after FillUp every GPU may hold a copy of the full column array
 * whose entries are {neighbour id, feature-cache slot of that neighbour} pairs.  The sampler's scattered pick reads the pair in
 * the one sector it fetches anyway, and the gather no longer looks the row's cache slot up (a 128-byte line per row for four
 * bytes: SS/cache/cache.cu:180-215 does it with a hash find per row).  Results are unchanged.  Returns 1 when GPU dev has it. */
int32_t legion_graph_column_slots(const LegionGraphStorage* g, int32_t dev);
/* The cached CSR logical GPU dev holds after a fill (GraphStorage::GraphCache, SS/storage/graph_storage.cu:76-111; kernels
 * SS/storage/graph_storage_impl.cuh:33-53): device pointers to int64 index[capacity + 1] and int32 dst[index[capacity]]; nulls before a
 * fill.  Introspection: tests/test_gpu_ref_graph_cache.py compares them with what the reference's own kernels produce (oracle/_ref). */
void legion_graph_cached_csr(const LegionGraphStorage* g, int32_t dev, const int64_t** index_out, const int32_t** dst_out);

/* FeatureStorage: SS/storage/feature_storage.cu:18-90.  ids/labels are HOST arrays copied to
 * device `dev_id`; mode selects the training / validation / testing set. */
LegionFeatureStorage* legion_feature_create(int32_t partition_count, int32_t total_num_nodes,
                                            int32_t float_feature_len,
                                            const float* all_float_feature_devptr);
void legion_feature_set_ids(LegionFeatureStorage* f, int32_t dev_id, int32_t mode,
                            const int32_t* host_ids, const int32_t* host_labels, int32_t count);
void legion_feature_destroy(LegionFeatureStorage* f);

/* MemoryPool + its buffers: SS/engine/server.cu:172-273 (GPURunner::Initialize buffer
 * allocation) and SS/engine/ipc_service.cu:134-211 (the 7 IPC-shared outputs per pipe slot).
 * fanout/hop_num size num_ids = B(1 + f1 + f1 f2 + ...) (SS/engine/server.cu:187-199). */
LegionMemoryPool* legion_pool_create(int32_t dev_id, int32_t total_num_nodes, int32_t batch_size,
                                     const int32_t* fanout, int32_t hop_num, int32_t float_feature_len,
                                     int32_t pipeline_depth);
/* SS/engine/server.cu:275-283: rows = int(1.2 * MaxIdNum) in the reference; caller passes rows. */
void legion_pool_alloc_features(LegionMemoryPool* p, int64_t rows);
void legion_pool_set_current_pipe(LegionMemoryPool* p, int32_t pipe);
void legion_pool_set_mode_iter(LegionMemoryPool* p, int32_t mode, int32_t iter);
int32_t legion_pool_num_ids(const LegionMemoryPool* p);
/* which: 0 sampled_ids 1 float_features 2 labels 3 agg_src_off 4 agg_dst_off 5 node_counter
 *        6 edge_counter (the IPC slot order, SS/engine/ipc_service.cu:163-169,203);
 *        7 agg_src_ids 8 agg_dst_ids 9 cache_search_buffer 10 tmp_part_ind 11 tmp_part_off
 *        12 position_map (always null here); 13 node_slot (new: int32[num_ids], the feature-cache slot the sampler carried for
 *        each node of the batch, -3 = not carried: the gather looks node_map up).  Returns the device pointer of the CURRENT pipe slot. */
void* legion_pool_buffer(LegionMemoryPool* p, int32_t which);
/* New in this build.  The reference keeps first touches in accessed_map (N bits, memset per batch) + position_map (N entries,
 * SS/engine/memorypool.cuh:120-135); a pool here keeps NOTHING per vertex: a hop's claims are de-duplicated bucket by bucket in
 * LDS (legion_core.h).  hash buckets per lane: 8, 16, 64 or 256 by the pool's largest hop; state_bytes: one hop's claim lists +
 * the known lists (scale with the batch, not with the graph).  which = 12 of legion_pool_buffer (position_map) returns NULL. */
int32_t legion_pool_lds_buckets(const LegionMemoryPool* p);
int64_t legion_pool_state_bytes(const LegionMemoryPool* p);
/* Sticky error bits raised on the device for this pool (0 = none): 1 a de-duplication bucket that fits no LDS table, 2 batch larger than the
 * feature buffer (gather stopped at its end; the reference overruns, SS/engine/server.cu:277), 4 internal.  The
 * word lives in host-visible memory: reading it after the batch completed needs no copy. */
int32_t legion_pool_error(const LegionMemoryPool* p);
void legion_pool_destroy(LegionMemoryPool* p);

/* UnifiedCache: SS/cache/cache.cu:295-321 (Initialize), :323-328 (InitializeCacheController). */
LegionUnifiedCache* legion_cache_create(int64_t cache_memory, int32_t float_feature_len,
                                        int32_t train_step, int32_t device_count,
                                        int32_t total_num_nodes);
void legion_cache_init_controller(LegionUnifiedCache* c, int32_t dev_id);
/* New in this build (no counterpart in the reference, whose GPUs had 16-80 GB): with caches striped over a clique of Kg
 * GPUs (cache_impl.cuh:89-109) every member may ALSO keep a private copy of the clique's hottest rows, as many as
 * `bytes` hold.  Lookup results (hit mask, global slot in cache_search_buffer) are unchanged; a hit whose hotness rank is
 * below the replica size is read from local HBM instead of a peer over xGMI.  Call before legion_cache_fill_up*. */
void legion_cache_set_replica_memory(LegionUnifiedCache* c, int64_t bytes);
int32_t legion_cache_replica_rows(const LegionUnifiedCache* c, int32_t dev_id);
/* Enables row-source statistics of the gathers on dev_id (Kg > 1) and returns the totals so far:
 * out2[0] rows read through a stripe pointer (own or peer), out2[1] rows read from the local replica. */
void legion_cache_gather_stats(LegionUnifiedCache* c, int32_t dev_id, uint64_t* out2);
/* same with out3[2] = the part of out3[0] that came from ANOTHER member's stripe (over xGMI between physical GPUs) */
void legion_cache_gather_stats3(LegionUnifiedCache* c, int32_t dev_id, uint64_t* out3);
/* pauses (0) / resumes (1) the counting: it costs the gather an atomic per hit row, so measurements count in an untimed pass */
void legion_cache_gather_stats_enable(LegionUnifiedCache* c, int32_t on);
/* SS/cache/cache.cu:360-443.  Hotness is summed over the clique on the clique leader through
 * peer pointers (one process, several GPUs); when `world_reduced` is non-zero the caller has
 * already all-reduced the counters across processes with RCCL and they are used as they are. */
void legion_cache_candidate_selection(LegionUnifiedCache* c, int32_t cache_agg_mode,
                                      LegionGraphStorage* graph, int32_t world_reduced);
/* The hotness all-reduce as the product's own RCCL call (collective.hip; reference: the leader's aggregate_access loop over peer
 * pointers, SS/cache/cache.cu:408-411,428-431).  Inside ONE server process (a thread per GPU) CandidateSelection issues it itself
 * over the clique's distinct physical GPUs (LegionTuning.hotness_reduce).  With one process per GPU the host program carries a
 * 128-byte unique id from rank 0 to every rank (any channel), each rank joins with the logical GPU it owns, and
 * legion_cache_allreduce_hotness sums GPU dev_id's two uint64[N] counter arrays in place over all ranks; then
 * candidate_selection(world_reduced = 1).  Returns: 1 / the world size on success, 0 on failure. */
int32_t legion_collective_unique_id(void* out128);
int32_t legion_collective_init_rank(const void* id128, int32_t world, int32_t rank, int32_t dev_id);
int32_t legion_collective_allreduce_u64(void* devptr, int64_t count, double* ms_out);
void legion_collective_destroy(void);
int32_t legion_cache_allreduce_hotness(LegionUnifiedCache* c, int32_t dev_id, double* ms_out);
/* how the last candidate selection summed the counters of dev_id's clique: 0 nothing to sum / taken as given, 1 the leader loop
 * over peer pointers, 2 an RCCL all-reduce issued by the library */
int32_t legion_cache_hotness_reduce_path(const LegionUnifiedCache* c, int32_t dev_id);
/* SS/cache/cache.cu:445-551; counters[2] = PCIe/xGMI transaction counts (zeros reproduce v2). */
void legion_cache_cost_model(LegionUnifiedCache* c, LegionFeatureStorage* feature,
                             LegionGraphStorage* graph, const uint64_t* counters, int32_t train_step);
/* Bypass for the all-resident configuration: fix the capacities instead of solving for them. */
void legion_cache_set_capacity(LegionUnifiedCache* c, int32_t node_capacity, int32_t edge_capacity);
/* SS/cache/cache.cu:553-611 */
void legion_cache_fill_up(LegionUnifiedCache* c, LegionFeatureStorage* feature, LegionGraphStorage* graph);
/* The hybrid CPU-cache / GPU-cache tier: UnifiedCache::HybridInit (SS/cache/cache.cu:614-670) with
 * PreSCCacheController::HybridInsert (:138-153, HybridInitPair SS/cache/cache_impl.cuh:113-123) INSTEAD OF candidate_selection +
 * cost_model + fill_up -- the call the reference keeps commented out at SS/engine/server.cu:112.  Every GPU orders the vertices by
 * its OWN PreSC counters (no clique sum); the gpu_cache_capacity hottest rows live in an HBM cache (slots cpu_cap + rank), the next
 * cpu_cache_capacity in a mapped pinned host cache (slots rank - gpu_cap), everything else is a miss; the topology maps stay empty.
 * The gather then resolves a slot as feat_cache_lookup does (cache_impl.cuh:202-235).  The capacities are the disk-mode
 * meta_config fields 14 and 15 (SS/storage/storage_management.cu:91-94).  The reference fills neither cache (cache.cu:616,656);
 * here both are filled from the feature table.  miss_from_table != 0: a miss row is read from the FeatureStorage table (the stand-in
 * for the unreleased SSD reader, IOSubmit SS/engine/operator_impl.cu:522-539); 0: as that kernel, a miss row is left unwritten. */
void legion_cache_hybrid_init(LegionUnifiedCache* c, LegionFeatureStorage* feature, LegionGraphStorage* graph,
                              int32_t cpu_cache_capacity, int32_t gpu_cache_capacity, int32_t miss_from_table);
/* device address of GPU dev_id's CPU cache (mapped pinned host memory, float[cpu_cache_capacity x D]) / of its HBM cache
 * (float[capacity x D]: this member's stripe after fill_up, the GPU cache after hybrid_init); null before either */
const float* legion_cache_hybrid_cpu_cache(const LegionUnifiedCache* c, int32_t dev_id);
const float* legion_cache_feature_cache(const LegionUnifiedCache* c, int32_t dev_id);
void legion_cache_destroy(LegionUnifiedCache* c);
/* A clique spread over PROCESSES (one process per GPU, Kg = world size): every rank owns member
 * `dev` = its rank (legion_set_local_device), builds its own stripe, publishes three IPC handles
 * (feature cache, cached CSR indptr, cached CSR columns; 192 bytes), opens the other members' handles
 * and links them into its pointer tables: remote cache rows and adjacency are then read with direct
 * peer loads over xGMI, as the reference does over NVLink inside one process (cache_impl.cuh:268,
 * operator_impl.cu:228-242).  Order on every rank: PreSC -> all-reduce hotness (RCCL) ->
 * set_peer_max_ids -> candidate_selection(world_reduced = 1) -> cost_model -> fill_up_local -> export ->
 * all-gather handles -> import_peer for every other rank -> fill_up_link. */
void legion_set_local_device(int32_t dev);
void legion_cache_set_peer_max_ids(LegionUnifiedCache* c, const int32_t* max_ids, int32_t n);
void legion_cache_fill_up_local(LegionUnifiedCache* c, LegionFeatureStorage* feature, LegionGraphStorage* graph);
void legion_cache_export(LegionUnifiedCache* c, LegionGraphStorage* graph, int32_t dev_id, void* handles192);
void legion_cache_import_peer(LegionUnifiedCache* c, LegionGraphStorage* graph, int32_t local_dev, int32_t peer_dev,
                              const void* handles192);
void legion_cache_fill_up_link(LegionUnifiedCache* c, LegionFeatureStorage* feature, LegionGraphStorage* graph);

/* introspection for the parity tests; arrays are device pointers on the clique leader */
int32_t legion_cache_node_capacity(const LegionUnifiedCache* c, int32_t dev_id);
int32_t legion_cache_edge_capacity(const LegionUnifiedCache* c, int32_t dev_id);
int32_t legion_cache_max_id_num(const LegionUnifiedCache* c, int32_t dev_id);
/* 64-byte transactions GPU dev_id's PreSC epoch spent on topology reads, counted by the sampler itself
 * (per sampled row: 1 for the row-pointer pair + min(fan-out, ceil(4*deg/64)) for the picks).  This is the
 * quantity the paper took from Intel PCM and v2 hard-wires to 0 (SS/engine/server.cu:105-110,
 * SS/engine/monitor.cuh); sum it over the GPUs and pass it as counters[0] to legion_cache_cost_model to
 * restore the topology-vs-feature trade-off, or pass {0,0} to reproduce v2. */
uint64_t legion_cache_topo_transactions(LegionUnifiedCache* c, int32_t dev_id);
/* which: 0 QF 1 QT (int32[N]) 2 AF 3 AT (uint64[N]) 4 node_access_time 5 edge_access_time
 *        (uint64[N], per device) 6 node_map 8 edge_offset_map (int32[N]) 7 edge_index_map (int8[N]) */
void* legion_cache_array(LegionUnifiedCache* c, int32_t dev_id, int32_t which);
/* UnifiedCache::FindTopo / FindFeat (SS/cache/cache.cu:335-357) on device arrays.  The sampler and
 * the gather resolve rows themselves (row headers, fused lookup); these remain for callers that
 * want the reference's explicit outputs: owner device / row offset or -2, cache slot or -2. */
void legion_cache_find_topo(LegionUnifiedCache* c, int32_t dev_id, legion_stream_t stream,
                            const int32_t* input_ids, int32_t batch_size, char* partition_index,
                            int32_t* partition_offset);
void legion_cache_find_feat(LegionUnifiedCache* c, int32_t dev_id, legion_stream_t stream,
                            const int32_t* sampled_ids, int32_t* cache_offset, const int32_t* node_counter,
                            int32_t op_id);

/* =====================================================================================
 * 3. Runner / Server / IPC -- SS/engine/server.h:5-33, SS/engine/ipc_service.h:6-35,
 *    sampling_server/sampling_server.cpp:7 (Run(fanout, gpu_number, in_memory_mode, cache_mode)).
 * ===================================================================================== */
LegionServer* NewGPUServer(void);
void legion_server_initialize(LegionServer* s, int32_t global_shard_count, const int32_t* fanout,
                              int32_t hop_num, int32_t in_memory_mode);
void legion_server_presc(LegionServer* s, int32_t cache_agg_mode);
void legion_server_run(LegionServer* s);
void legion_server_finalize(LegionServer* s);
/* pybind `sampling_server.Run` equivalent; reads ./meta_config from the cwd */
int32_t legion_run(const int32_t* fanout, int32_t hop_num, int32_t gpu_number, int32_t in_memory_mode,
                   int32_t cache_mode);

LegionIPCEnv* NewIPCEnv(int32_t device_count);
/* step arithmetic, SS/engine/ipc_service.cu:60-132,213-253 (host only, no GPU needed) */
void legion_ipc_coordinate(LegionIPCEnv* e, int32_t partition_count, const int32_t* train_num,
                           const int32_t* valid_num, const int32_t* test_num, int32_t raw_batch_size,
                           int32_t epoch);
int32_t legion_ipc_train_step(LegionIPCEnv* e);
int32_t legion_ipc_max_step(LegionIPCEnv* e);
int32_t legion_ipc_current_mode(LegionIPCEnv* e, int32_t global_batch_id);
int32_t legion_ipc_local_batch_id(LegionIPCEnv* e, int32_t global_batch_id);
int32_t legion_ipc_current_batchsize(LegionIPCEnv* e, int32_t dev_id, int32_t mode);
void legion_ipc_finalize(LegionIPCEnv* e);

/* Lane groups: every kernel of the path takes an array of per-mini-batch buffer descriptors and is
 * launched with grid.y = lanes, so ONE launch of each kernel serves `n` independent mini-batches
 * (lane i produces batch counter0 + i into pool i).  The reference-shaped operators above are the
 * n = 1 case.  No reference counterpart: this is how the path keeps 256 CUs busy at B = 1024. */
typedef struct LegionLaneGroup LegionLaneGroup;
LegionLaneGroup* legion_group_create(LegionMemoryPool** pools, int32_t n);
void legion_group_set_iter_state(LegionLaneGroup* g, int32_t* iter_state_devptr);
void legion_group_destroy(LegionLaneGroup* g);
void legion_enqueue_group(legion_stream_t strm_hdl, LegionGraphStorage* graph, LegionFeatureStorage* feature,
                          LegionUnifiedCache* cache, LegionLaneGroup* group, int32_t batch_size,
                          int32_t counter0, int32_t dev_id, int32_t mode, const int32_t* fanout, int32_t hop_num);
/* same with only the first n_active lanes working */
void legion_enqueue_group_n(legion_stream_t strm_hdl, LegionGraphStorage* graph, LegionFeatureStorage* feature,
                            LegionUnifiedCache* cache, LegionLaneGroup* group, int32_t n_active, int32_t batch_size,
                            int32_t counter0, int32_t dev_id, int32_t mode, const int32_t* fanout, int32_t hop_num);

/* Pipeline: `slots` groups of `group_size` mini-batches in flight on one GPU; each group is replayed
 * as one hipGraph on its own stream (sizes and the batch index live on the device, so a replay
 * needs no argument update).  The MI355X-native form of the Runner's inter-batch pipe
 * (SS/engine/server.cu:302-332 with INTERBATCH_CON output slots): submit() never blocks on the
 * group it enqueues, only on the slot's previous one.  feature_rows sizes each lane's feature buffer
 * (SS/engine/server.cu:275-283).  use_graph: bit 0 = replay hipGraphs (else eager launches), bit 1 =
 * let kernels of different slots overlap on the GPU (default: slots are chained by an event, so a
 * group's launch latency hides behind the previous group but their kernels never share the GPU). */
typedef struct LegionPipeline LegionPipeline;
LegionPipeline* legion_pipeline_create(LegionGraphStorage* graph, LegionFeatureStorage* feature,
                                       LegionUnifiedCache* cache, int32_t dev_id, int32_t batch_size,
                                       const int32_t* fanout, int32_t hop_num, int32_t group_size,
                                       int32_t slots, int64_t feature_rows, int32_t use_graph);
/* enqueues batches counter0 .. counter0 + group_size - 1; returns the slot */
int32_t legion_pipeline_submit(LegionPipeline* p, int32_t counter0, int32_t mode);
/* only the first n_active lanes work (tail of a run that is not a multiple of group_size) */
int32_t legion_pipeline_submit_n(LegionPipeline* p, int32_t counter0, int32_t mode, int32_t n_active);
void legion_pipeline_wait(LegionPipeline* p, int32_t slot);                          /* slot < 0: all slots */
LegionMemoryPool* legion_pipeline_pool(LegionPipeline* p, int32_t slot, int32_t lane);
void legion_pipeline_destroy(LegionPipeline* p);
/* Owner-bucketed bulk transfer for a striped feature cache (LegionTuning.peer_gather = bulk; SURVEY section 7 "hard parts"; the
 * alternative to the 512-1024-byte direct peer loads of SS/cache/cache_impl.cuh:268).  A pipeline created with use_graph bits 5 + 6 keeps
 * its lanes' trainer-visible arrays in ONE arena other GPUs and processes can reach (shuffled physical chunks created exportable; a
 * plain allocation with LegionTuning.arena_scatter_mb = 0).  Per group, phase A on every member of the clique
 * (sampler, bucket pass that lists per owner {row inside its stripe, destination inside the requester's arena}, gather of everything
 * that is not another member's stripe) -> the caller's barrier -> phase B on every member as an OWNER (its rows, read from its own
 * HBM, pushed to the requesters with coalesced posted stores) -> barrier.  Lookup results and rows are those of the direct
 * arrangement, bit for bit.  _export / _import carry what a member in another process needs (<= 512 bytes: IPC handles of the lists and
 * of a plain arena, or the name of the abstract unix socket that serves a chunked arena's file descriptors to a same-user peer), _link
 * takes a member that lives in the same process (and grants its GPU access to a chunked arena). */
int32_t legion_pipeline_bulk_enable(LegionPipeline* p);
int32_t legion_pipeline_bulk_export(LegionPipeline* p, void* out_handles, int32_t out_bytes);
int32_t legion_pipeline_bulk_import(LegionPipeline* p, const void* handles);
int32_t legion_pipeline_bulk_link(LegionPipeline* p, LegionPipeline* other);
int32_t legion_pipeline_bulk_phase_a(LegionPipeline* p, int32_t counter0, int32_t mode, int32_t n_active, int32_t batch_size);
void legion_pipeline_bulk_phase_b(LegionPipeline* p, int32_t slot);
int64_t legion_pipeline_bulk_listed(LegionPipeline* p, int32_t slot);
/* Measurement aid: HIP events on each slot's stream around every gather launch.  While it is on,
 * groups are launched eagerly (HIP cannot time events recorded by graph nodes).  read() fills, per
 * gather op id, the summed elapsed ms and the launch count of every batch waited for since begin();
 * returns the op count. */
void legion_pipeline_profile_begin(LegionPipeline* p);
void legion_pipeline_profile_end(LegionPipeline* p);
int32_t legion_pipeline_profile_read(LegionPipeline* p, int32_t* op_ids, double* ms_sums, int64_t* counts,
                                     int32_t cap);
/* Measurement aid: the gather of the LAST op of the group sitting in `slot`, launched `repeats` more times over the lanes as they
 * stand -- the kernel instance, grid and ranges of the group's own op list (multiGPU_feat_cache_lookup for op 3H+1,
 * SS/cache/cache_impl.cuh:239-272) -- each launch between two HIP events on the slot's stream; ms_each[i] = its duration.  A caller
 * may rewrite the lanes' ids (legion_pool_buffer 0 / 13) in between: bench.py's roofline.cold gathers rows of which none repeats
 * inside the launch.  Returns the launches timed. */
int32_t legion_pipeline_regather_last(LegionPipeline* p, int32_t slot, int32_t n_active, int32_t repeats, double* ms_each);

/* =====================================================================================
 * 4. Kernel-level launchers (what the operators call), exported so the hot kernels can be
 *    measured and tested alone.
 * ===================================================================================== */
/* The gather of SS/cache/cache_impl.cuh:239-272 with the id->slot lookup of
 * SS/cache/cache.cu:180-215 fused in.  range[0..1] = {row offset, row count} is read on the
 * device (node_counter layout).  node_map may be NULL (every row is a miss). */
void legion_gather_rows(legion_stream_t stream, const float* full_table, const float* const* cache_tables,
                        const int32_t* node_map, int32_t node_capacity, int32_t float_feature_len,
                        int32_t total_num_nodes, const int32_t* sampled_ids, int32_t* cache_index_out,
                        const int32_t* range_devptr, float* dst, int32_t max_rows);
/* the draw of SS/engine/operator_impl.cu:235-238 evaluated on the GPU for n (idx, deg) pairs */
void legion_draw_batch(legion_stream_t stream, const int32_t* idx, const int32_t* deg, int32_t* out,
                       int32_t n);
/* Measurement aid (no reference counterpart): while enabled, FeatureCacheLookup records a HIP event
 * on its own stream before and after the gather launch.  _end returns how many gathers were timed
 * and fills their elapsed ms and op ids; call it after synchronising the stream. */
void legion_pool_profile_begin(LegionMemoryPool* p, int32_t max_ops);
int32_t legion_pool_profile_end(LegionMemoryPool* p, float* out_ms, int32_t* out_op, int32_t cap);

/* Cumulative PCIe / xGMI byte counters of logical GPU dev_id from the driver's gpu_metrics table (the MI355X counterpart
 * of the Intel-PCM PCIe counters that feed CostModel in the paper: SS/engine/server.cu:105-110, SS/engine/monitor.cuh).
 * Returns 1 and fills the two totals, or 0 when the table is missing / has an unknown revision.  Host-only. */
int32_t legion_link_counters(int32_t dev_id, uint64_t* pcie_bytes, uint64_t* xgmi_bytes);
/* The same table in full: PCIe total, xGMI bytes READ and WRITTEN by this GPU in total and per link (8 links; 7 populated
 * on an 8-GPU MI355X node), the table revision that was found (known: 1.8) and the GPU's PCI bus id.  CostModel's second
 * counter (counters[1]) is fed with the xGMI bytes the clique's members READ during the PreSC epoch / 64
 * (SS/engine/server.cu:105-106: the two PCM counters are summed into "transactions of topology", cache.cu:459). */
typedef struct LegionLinkCounters {
    uint64_t pcie_bytes;
    uint64_t xgmi_read_bytes, xgmi_write_bytes;
    uint64_t xgmi_read_bytes_link[8], xgmi_write_bytes_link[8];
    int32_t format_revision, content_revision;
    char pci_bus_id[32];
    int32_t source;              /* 1 rocm_smi_lib's versioned decoder (rsmi_dev_gpu_metrics_info_get), 2 the table parsed by byte offset */
    int32_t reserved;
} LegionLinkCounters;
int32_t legion_link_counters_ex(int32_t dev_id, LegionLinkCounters* out);
/* the same from ONE source (1 / 2 as above; 0 = the library's decoder first, the byte-offset parser second, as _ex does) */
int32_t legion_link_counters_from(int32_t dev_id, int32_t source, LegionLinkCounters* out);
/* 64-byte transactions the gathers of dev_id have so far read from OTHER members' stripes of a striped feature cache
 * (rows read through a peer's pointer x row bytes / 64; enables the row-source statistics like
 * legion_cache_gather_stats).  The computed stand-in for the xGMI counter where the driver's table is unavailable or
 * cannot move (one physical GPU). */
uint64_t legion_cache_peer_transactions(LegionUnifiedCache* c, int32_t dev_id);

/* =====================================================================================
 * 5. Synthetic workload generators (RMAT + counter-hash features); device side.
 * ===================================================================================== */
void legion_synth_rmat_edges(legion_stream_t stream, int32_t scale, int64_t num_edges, uint64_t seed,
                             int32_t* src_out, int32_t* dst_out);
/* same edges with Graph500-style label scrambling (a keyed bijection of [0, 2^scale)); key 0 = none */
void legion_synth_rmat_edges_scrambled(legion_stream_t stream, int32_t scale, int64_t num_edges, uint64_t seed,
                                       int32_t* src_out, int32_t* dst_out, uint64_t scramble_key);
void legion_synth_features(legion_stream_t stream, float* out, int64_t first_row, int64_t num_rows,
                           int32_t dim, uint64_t seed);
void legion_synth_feature_check(legion_stream_t stream, const float* rows, const int32_t* ids,
                                int64_t num_rows, int32_t dim, uint64_t seed,
                                unsigned long long* mismatch_count_devptr);

/* Measurement aid (no reference counterpart): one launch that loads every float of a batch's feature rows and every entry of its
 * two COO arrays and folds them into *acc_devptr -- a trainer-side consumer that really reads its batches
 * (tools/server_throughput.py --consume, bench.py's boundary.consuming_trainer). */
void legion_consume_batch(legion_stream_t stream, const float* feats, int64_t n_floats, const int32_t* src, const int32_t* dst,
                          int64_t n_edges, double* acc_devptr);

/* Spill-over tier: mapped pinned host memory (what the reference uses for the full CSR / feature table,
 * SS/storage/storage_management.cu:106-107,161).  Returns the pointer the GPU dereferences; *host_ptr_out
 * is the host-side address to fill and to pass to legion_host_free. */
void* legion_host_alloc(int64_t num_bytes, void** host_ptr_out);
void legion_host_free(void* host_ptr);

/* One process per GPU: logical GPU d of this process is physical GPU (base + d) % device_count.
 * Call once before creating any object (bench.py passes LOCAL_RANK). */
void legion_set_device_base(int32_t base);
int32_t legion_get_device_base(void);

/* =====================================================================================
 * 6. Tuning.  Every switch that changes how the path runs (never what it computes) lives in ONE
 *    struct.  The library keeps one process-wide copy; it is (re)filled from the LEGION_* environment
 *    variables named below by legion_tuning_from_env(), which the library itself calls whenever a
 *    MemoryPool, a Pipeline or a Server is created -- never inside a launch path.  A host program may
 *    instead fill the struct and call legion_tuning_set(): values set that way are kept (the
 *    environment is then only read again after legion_tuning_from_env() is called explicitly).
 *    No reference counterpart (the reference has compile-time constants only, system_config.cuh).
 *    Not tuning and therefore still plain environment: LEGION_IPC_NAMESPACE, LEGION_IPC_LOCAL,
 *    LEGION_IPC_DEVICE (deployment: which shm names / which GPU a trainer attaches to), and the trainer module's own
 *    LEGION_NO_DIRECT_VIEWS / LEGION_NO_SHM_MIRROR (ipc_service is a separate extension: it reads them once, in initialize()).
 * ===================================================================================== */
typedef struct LegionTuning {
    /* -- sampler ------------------------------------------------------------------------------------------------------------ */
    int32_t lds_small_buckets;   /* LEGION_LDS_SMALL_BUCKETS (0 auto | 8 | 16): hash buckets per lane of pools whose hops have <= 2^19 slots;
                                    auto = 16 where PreSC saw more last-hop edges + earlier nodes than 8 buckets take in one pass */
    int32_t lds_part_wg;         /* LEGION_LDS_PART_WG     (8192): workgroups a sampling launch of the 64/256-bucket classes aims for (picks the
                                    partition tile: 1..8 super tiles) */
    int32_t sample_max_wg;       /* LEGION_SAMPLE_MAX_WG   (4096): workgroup cap of the strided sampler grids */
    int32_t lds_known_cap;       /* LEGION_LDS_KNOWN_CAP   (0 = 2 x an even share): entries per known-node list (tests force the fallback) */
    int32_t lds_claim_cap;       /* LEGION_LDS_CLAIM_CAP   (0 = 2 x an even share): entries per claim list (tests force the fallback) */
    int32_t col_slots;           /* LEGION_COL_SLOTS       (-1 auto): the {neighbour id, feature-cache slot} copy of the column array that
                                    lets the gather skip its node_map lookup (8 B per edge of HBM per GPU): 1 always, 0 never,
                                    -1 when the column array is device memory and the copy fits half of the HBM that is free after the fills, leaving 24 GB */
    /* -- gather ------------------------------------------------------------------------------------------------------------- */
    int32_t gather_rows_per_wg;  /* LEGION_GATHER_ROWS     (0 = by row width): rows per gather workgroup, 16|32|64|128|256 */
    int32_t peer_gather;         /* LEGION_PEER_GATHER=direct|bulk -> 0|1: rows of OTHER members' stripes of a striped feature cache are
                                    read by direct peer loads, or pushed by their owners in bulk (pipeline.hip) */
    /* -- launch groups ------------------------------------------------------------------------------------------------------- */
    int32_t arena_scatter_mb;    /* LEGION_ARENA_SCATTER_MB (2; 0 = plain allocations): lane arenas are built from physical chunks of this many MB
                                    mapped in shuffled order (HIP virtual memory management): the gathers write a group's rows all over the
                                    HBM instead of into one contiguous range (0.87 instead of 0.80 of the peak) */
    int32_t weave_priority;      /* LEGION_WEAVE_PRIORITY  (-1): priority of the light stream (heads of the next group): -1 low, 0 equal, 1 high */
    int32_t markers;             /* LEGION_MARKERS         (1): roctx ranges around ops and launch groups (visible to rocprofv3 --marker-trace) */
    /* -- Runner (server side of the boundary) -------------------------------------------------------------------------------- */
    int32_t runner_graph;        /* LEGION_RUNNER_GRAPH    (1): Runner serves from lane groups + hipGraph; 0 = operator by operator */
    int32_t runner_lanes;        /* LEGION_RUNNER_LANES    (0 = min(512, 524288 / batch)): lanes of a Runner group */
    int32_t runner_slots;        /* LEGION_RUNNER_SLOTS    (3): launch groups the Runner keeps in flight (2..4): one being handed over, one
                                    running, one queued behind it */
    int32_t runner_handover;     /* LEGION_RUNNER_HANDOVER=auto|gather -> 0|1: how the Runner's batches reach a trainer.  auto: a trainer end that
                                    opened the lane arena gets a batch as VIEWS of its lane (no per-batch GPU work), any other gets its rows
                                    gathered straight into the pipe slot by one launch per batch.  gather: that for every trainer end */
    int32_t runner_ho_stream;    /* LEGION_RUNNER_HO_STREAM (2): hand-over streams of the gather hand-over: 0 the sampler's, 1 one shared, 2 one per pipe slot */
    int32_t runner_spin_us;      /* LEGION_RUNNER_SPIN_US  (-1 auto): how long the Runner polls (a trainer's semaphore, a group's completion) before it
                                    blocks.  auto: 20 us when batches are handed over as views (a group completes every few ms: the host
                                    sleeps in between), polling only with the gather hand-over (per-batch latency is the rate there) */
    int32_t runner_overflow;     /* LEGION_RUNNER_OVERFLOW (1): a batch with more rows than 1.2 x the PreSC maximum (the lanes' feature buffers) still goes
                                    out whole: views from one of two num_ids-row overflow buffers inside the arena, the other hand-overs into pipe-slot
                                    buffers of num_ids rows; 0 = the rule everywhere: a views server stops there, the slab path truncates with a warning */
    int32_t runner_stats;        /* LEGION_RUNNER_STATS    (0): print where a hand-over's time went at Finalize */
    int32_t shm_mirror;          /* LEGION_NO_SHM_MIRROR unset -> 1: counters also go to a host-visible mirror (no D2H copy per batch) */
    /* -- set-up -------------------------------------------------------------------------------------------------------------- */
    int32_t table_placement;     /* LEGION_TABLE_PLACEMENT=hbm|pinned -> 0|1: where the server puts the full CSR / feature table */
    int32_t hotness_reduce;      /* LEGION_HOTNESS_REDUCE=auto|p2p|rccl -> -1|0|1: the clique sum of the access counters: rccl = all-reduce
                                    over the server's distinct physical GPUs, p2p = the reference's leader loop over peer pointers
                                    (SS/cache/cache.cu:408-411); auto = rccl when the members sit on distinct physical GPUs, p2p if that fails */
    int32_t link_counters;       /* LEGION_LINK_COUNTERS=v2|measured|smi|"a,b" -> 0|1|2|3: what feeds CostModel's counters */
    uint64_t link_counter_values[2];   /* the injected pair of LEGION_LINK_COUNTERS="a,b" */
} LegionTuning;
void legion_tuning_from_env(void);
void legion_tuning_get(LegionTuning* out);
void legion_tuning_set(const LegionTuning* in);

/* library / device info */
const char* legion_version(void);
int32_t legion_device_count(void);

#ifdef __cplusplus
}
#endif
#endif /* LEGION_HIP_H */
