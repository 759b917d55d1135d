// operators.hip -- the five extern "C" operator entry points and the Operator plug-in classes.
//
// Reference: SS/engine/operator_impl.cuh:11-63 (declarations), SS/engine/operator_impl.cu:92-172
// (BatchGenerate), :401-499 (RandomSample), :502-519 (FeatureCacheLookup), :522-539 (IOSubmit),
// :551-580 (IOComplete); SS/engine/operator.cu:15-122 (the five Operator::run bodies).
//
// Same names, argument order, null-pointer behaviour (message + return) and error convention.
// Differences a maintainer should know about:
//   * nothing here blocks on the device: the reference's 64-byte counter read-backs
//     (operator_impl.cu:439-445, cache.cu:187-188) are gone because every kernel reads the
//     frontier / node range from the counters in device memory;
//   * FindTopo runs inside the sampling kernel and FindFeat inside the gather kernel; their
//     outputs (tmp_part_ind/tmp_part_off, cache_search_buffer) are still written;
//   * counter_update is folded into the kernels that produce the counts;
//   * the accessed bitmap is fused into the position map (see legion_core.h), so BatchGenerate
//     does not memset N/8 bytes per batch and IOComplete restores the map in every mode.
#include "legion_core.h"

#include <iostream>

extern "C" void BatchGenerate(legion_stream_t strm_hdl, LegionFeatureStorage* feature_,
                              LegionUnifiedCache* cache_, LegionMemoryPool* memorypool_,
                              int32_t batch_size, int32_t counter, int32_t part_id, int32_t dev_id,
                              int32_t mode, bool is_presc, int32_t hop_num);
extern "C" void RandomSample(legion_stream_t strm_hdl, LegionGraphStorage* graph_, LegionUnifiedCache* cache_,
                             LegionMemoryPool* memorypool_, int32_t count, int32_t dev_id, int32_t op_id,
                             bool is_presc);

// The C handles are the C++ objects themselves, except the cache, whose handle boxes the
// UnifiedCache as its first member (cache.hip: LegionCacheBox).
static inline UnifiedCache* cache_of(LegionUnifiedCache* c) { return reinterpret_cast<UnifiedCache*>(c); }

// ---- lane-group bodies: every operator works on n lanes (n = 1 for the reference-shaped calls) ----
struct LegionLaneGroup {
    std::vector<MemoryPool*> pools;
    LanePtrs* d_lanes = nullptr;      // contiguous device copy of every pool's current lane
    int32_t* iter_state = nullptr;    // device {next iteration of lane 0, stride} for graph replay, or null
};

static bool seed_set(FeatureStorage* feature, int32_t dev_id, int32_t mode, int32_t*& all_ids, int32_t*& all_labels,
                     int32_t& total_cap)
{
    all_ids = nullptr;
    all_labels = nullptr;
    total_cap = 0;
    if (mode == TRAINMODE) {
        all_ids = feature->GetTrainingSetIds(dev_id);
        all_labels = feature->GetTrainingLabels(dev_id);
        total_cap = feature->TrainingSetSize(dev_id);
    } else if (mode == VALIDMODE) {
        all_ids = feature->GetValidationSetIds(dev_id);
        all_labels = feature->GetValidationLabels(dev_id);
        total_cap = feature->ValidationSetSize(dev_id);
    } else if (mode == TESTMODE) {
        all_ids = feature->GetTestingSetIds(dev_id);
        all_labels = feature->GetTestingLabels(dev_id);
        total_cap = feature->TestingSetSize(dev_id);
    } else {
        std::cout << "invalid mode: " << mode << "\n";
    }
    if (all_ids == nullptr) {
        std::cout << "invalid src id ptr\n";
        return false;
    }
    if (all_labels == nullptr) {
        std::cout << "invalid label ptr\n";
        return false;
    }
    return true;
}

static void do_batch_generate(hipStream_t s, FeatureStorage* feature, const LanePtrs* d_lanes, int32_t n_lanes,
                              MemoryPool* pool0, int32_t batch_size, int32_t counter, int32_t dev_id, int32_t mode,
                              int32_t hop_num, const int32_t* iter_state)
{
    lg::Range mark("op0 batch_generate lanes=%d B=%d", n_lanes, batch_size);
    lg::SeedParams p;
    int32_t* all_ids = nullptr;
    int32_t* all_labels = nullptr;
    if (!seed_set(feature, dev_id, mode, all_ids, all_labels, p.total_cap)) return;
    p.all_ids = all_ids;
    p.all_labels = all_labels;
    if (batch_size > pool0->batch_size) {
        std::cout << "batch size " << batch_size << " exceeds the pool's " << pool0->batch_size << "\n";
        return;
    }
    p.batch_size = batch_size;
    p.counter0 = counter;
    p.hop_num = hop_num;
    p.iter_state = iter_state;
    lg::launch_batch_generate(s, p, d_lanes, n_lanes);
    // cache->FindFeat(op 0) (operator_impl.cu:167-170) happens inside FeatureCacheLookup(op 1)
}

static void do_random_sample(hipStream_t s, GraphStorage* graph, UnifiedCache* cache, const LanePtrs* d_lanes,
                             int32_t n_lanes, MemoryPool* pool0, int32_t count, int32_t dev_id, int32_t op_id,
                             bool is_presc)
{
    if (op_id < INTRABATCH_CON || op_id % INTRABATCH_CON != 0 || count < 1) {
        printf("Sampling Parameters Error\n");   // counter_update's complaint, operator_impl.cu:86-88
        return;
    }
    lg::Range mark("op%d sample%s fanout=%d lanes=%d", op_id, is_presc ? " (presc)" : "", count, n_lanes);
    lg::HopParams p;
    p.op_id = op_id;
    p.count = count;
    p.partition_count = graph->GetPartitionCount();
    p.csr_dst_node_ids = graph->GetCSRNodeMatrix(dev_id);
    // column slots only from the fill of THIS cache that built them (legion_core.h GraphStorage::BuildColumnSlots)
    const bool pairs_ok = !is_presc && pool0->slot_fs != nullptr && cache != nullptr && graph->ColumnSlotsStamp(dev_id) != 0 &&
                          graph->ColumnSlotsStamp(dev_id) == cache->FillStamp();
    p.csr_dst_x = pairs_ok ? graph->GetCSRXMatrix(dev_id) : nullptr;
    p.col_full = graph->GetCSRNodeMatrixCPU();
    p.colx_full = p.csr_dst_x != nullptr ? graph->GetColumnSlotsFull(dev_id) : nullptr;
    p.row_hdr = graph->GetRowHeaders(dev_id);
    p.last_hop = (size_t)(op_id / INTRABATCH_CON) + 1 >= pool0->max_new.size();
    p.is_presc = is_presc;
    const size_t hop = (size_t)(op_id / INTRABATCH_CON);          // slots of this hop <= B f1..fh
    p.max_slots = (int32_t)(hop < pool0->max_new.size() ? pool0->max_new[hop] : pool0->max_slots);
    p.edge_access_time = (is_presc && cache) ? cache->GetEdgeAccessedMap(dev_id) : nullptr;   // :473
    p.topo_transactions = (is_presc && cache) ? cache->Controller(dev_id)->GetTopoTransactions() : nullptr;
    p.lds_bucket_bits = pool0->lds_bucket_bits;
    p.lds_k = 1;                             // (launch_random_sample picks the hop's partition tile)
    // 64- and 256-bucket classes, last hop: how many claims a de-duplication thread keeps in registers follows what PreSC saw in that hop
    // (+10 %: buckets are not even; a bucket that still outgrows its workgroup's registers re-reads its list, sweep by sweep)
    p.dedup_claims = LG_DEDUP_CLAIMS;
    if (p.last_hop && (pool0->lds_bucket_bits == LG_LDS_BITS_MEDIUM || pool0->lds_bucket_bits == LG_LDS_BITS_LARGE)) {
        const int64_t per_bucket = (pool0->last_hop_claims_hint * 11 / 10) >> pool0->lds_bucket_bits;
        if (per_bucket > (int64_t)LG_DEDUP_CLAIMS_MID * 1024 && pool0->lds_bucket_bits == LG_LDS_BITS_MEDIUM) p.dedup_claims = LG_DEDUP_CLAIMS_BIG;
        else if (per_bucket > (int64_t)LG_DEDUP_CLAIMS * 1024) p.dedup_claims = LG_DEDUP_CLAIMS_MID;
    }
    lg::launch_random_sample(s, p, d_lanes, n_lanes);
}

static void do_feature_lookup(hipStream_t s, UnifiedCache* cache, const LanePtrs* d_lanes, int32_t n_lanes,
                              MemoryPool* pool0, int32_t op_id, int32_t dev_id, bool use_snapshot, int32_t first_op_id = -1)
{
    if (pool0->GetFloatFeatures() == nullptr) {
        std::cout << "feature buffer not initialized\n";
        return;
    }
    if (cache->FeatureTable() == nullptr) {   // bound by FillUp (cache.cu:581-582) or legion_enqueue_batch
        std::cout << "invalid feature table ptr\n";
        return;
    }
    lg::Range mark("op%d gather lanes=%d first_op=%d", op_id, n_lanes, first_op_id);
    int64_t max_rows = pool0->feature_rows;
    if (max_rows > pool0->num_ids) max_rows = pool0->num_ids;
    const size_t hop = (size_t)(op_id / INTRABATCH_CON);          // grid bound: new nodes of op 3h <= B f1..fh
    int64_t bound = 0;                                            // + the earlier ops that ride along
    for (size_t h = (use_snapshot && first_op_id >= 0 && first_op_id < op_id) ? (size_t)(first_op_id / INTRABATCH_CON) : hop;
         h <= hop && h < pool0->max_new.size(); h++)
        bound += pool0->max_new[h];
    if (hop < pool0->max_new.size() && bound < max_rows) max_rows = bound;
    MemoryPool* pp = pool0;
    const bool prof = pp->prof_on && (size_t)(2 * pp->prof_used + 1) < pp->prof_events.size();
    if (prof) HIP_CALL(hipEventRecord(pp->prof_events[2 * pp->prof_used], s));
    cache->FeatCacheLookup(d_lanes, n_lanes, op_id, dev_id, s, (int32_t)max_rows, use_snapshot, first_op_id,
                           hop + 1 >= pool0->max_new.size(), false, (int32_t)std::min<int64_t>(pool0->grid_rows_hint, max_rows));
    if (prof) {
        HIP_CALL(hipEventRecord(pp->prof_events[2 * pp->prof_used + 1], s));
        pp->prof_op[pp->prof_used] = op_id;
        pp->prof_used++;
    }
}

extern "C" void BatchGenerate(legion_stream_t strm_hdl, LegionFeatureStorage* feature_,
                              LegionUnifiedCache* cache_, LegionMemoryPool* memorypool_,
                              int32_t batch_size, int32_t counter, int32_t part_id, int32_t dev_id,
                              int32_t mode, bool is_presc, int32_t hop_num)
{
    (void)part_id; (void)cache_; (void)is_presc;
    FeatureStorage* feature = reinterpret_cast<FeatureStorage*>(feature_);
    MemoryPool* memorypool = reinterpret_cast<MemoryPool*>(memorypool_);
    if (feature == nullptr || memorypool == nullptr) {
        std::cout << "invalid storage ptr\n";
        return;
    }
    do_batch_generate(static_cast<hipStream_t>(strm_hdl), feature, memorypool->DeviceLane(), 1, memorypool,
                      batch_size, counter, dev_id, mode, hop_num, memorypool->iter_state);
}

extern "C" void RandomSample(legion_stream_t strm_hdl, LegionGraphStorage* graph_, LegionUnifiedCache* cache_,
                             LegionMemoryPool* memorypool_, int32_t count, int32_t dev_id, int32_t op_id,
                             bool is_presc)
{
    GraphStorage* graph = reinterpret_cast<GraphStorage*>(graph_);
    MemoryPool* memorypool = reinterpret_cast<MemoryPool*>(memorypool_);
    if (graph == nullptr || memorypool == nullptr) {
        std::cout << "invalid storage ptr\n";
        return;
    }
    do_random_sample(static_cast<hipStream_t>(strm_hdl), graph, cache_of(cache_), memorypool->DeviceLane(), 1,
                     memorypool, count, dev_id, op_id, is_presc);
}

extern "C" void FeatureCacheLookup(legion_stream_t strm_hdl, LegionUnifiedCache* cache_,
                                   LegionMemoryPool* memorypool_, int32_t op_id, int32_t dev_id)
{
    MemoryPool* memorypool = reinterpret_cast<MemoryPool*>(memorypool_);
    UnifiedCache* cache = cache_of(cache_);
    if (cache == nullptr || memorypool == nullptr) {
        std::cout << "invalid storage ptr\n";
        return;
    }
    // reference semantics: gather the CURRENT new-node range (node_counter[0..1], counter_update(op%3==1))
    do_feature_lookup(static_cast<hipStream_t>(strm_hdl), cache, memorypool->DeviceLane(), 1, memorypool, op_id,
                      dev_id, false);
}

extern "C" void IOSubmit(legion_stream_t, LegionFeatureStorage*, LegionMemoryPool*, int32_t, int32_t)
{
    // SSD tier: the reference body is commented out (operator_impl.cu:522-539); nothing to do
}

extern "C" void IOComplete(legion_stream_t strm_hdl, LegionUnifiedCache* cache_, LegionMemoryPool* memorypool_,
                           int32_t dev_id, int32_t mode)
{
    MemoryPool* memorypool = reinterpret_cast<MemoryPool*>(memorypool_);
    UnifiedCache* cache = cache_of(cache_);
    if (memorypool == nullptr) {
        std::cout << "invalid storage ptr\n";
        return;
    }
    hipStream_t s = static_cast<hipStream_t>(strm_hdl);
    if (mode == TRAINMODE && cache != nullptr)   // CacheProfiling only in train mode (:558,:578)
        cache->CacheProfiling(memorypool->GetSampledIds(), memorypool->GetAggSrcId(), memorypool->GetAggDstId(),
                              memorypool->GetAggSrcOf(), memorypool->GetAggDstOf(), memorypool->GetNodeCounter(),
                              memorypool->GetEdgeCounter(), s, dev_id);
    lg::launch_end_of_batch(s, memorypool->DeviceLane(), 1, memorypool->iter_state);
}

// =============================================================================================
// SS/engine/operator.cu:15-122
class BatchGenerateOP : public Operator {
public:
    explicit BatchGenerateOP(int op_id) : op_id_(op_id) {}
    void run(OpParams* params) override
    {
        MemoryPool* memorypool = (MemoryPool*)(params->memorypool);
        IPCEnv* env = (IPCEnv*)(params->env);
        const int32_t device_id = params->device_id;
        const int32_t mode = memorypool->GetCurrentMode();
        const int32_t iter = memorypool->GetIter();
        const int32_t batch_size = env->GetCurrentBatchsize(device_id, mode);
        BatchGenerate(params->stream, (LegionFeatureStorage*)params->feature, (LegionUnifiedCache*)params->cache,
                      (LegionMemoryPool*)memorypool, batch_size, iter, device_id, device_id, mode,
                      params->is_presc, params->hop_num);
        HIP_CALL(hipEventRecord(params->event, params->stream));
    }
private:
    int op_id_;
};
Operator* NewBatchGenerateOP(int op_id) { return new BatchGenerateOP(op_id); }

class RandomSampleOP : public Operator {
public:
    explicit RandomSampleOP(int op_id) : op_id_(op_id) {}
    void run(OpParams* params) override
    {
        RandomSample(params->stream, (LegionGraphStorage*)params->graph, (LegionUnifiedCache*)params->cache,
                     (LegionMemoryPool*)params->memorypool, params->neighbor_count, params->device_id, op_id_,
                     params->is_presc);
        HIP_CALL(hipEventRecord(params->event, params->stream));
    }
private:
    int op_id_;
};
Operator* NewRandomSampleOP(int op_id) { return new RandomSampleOP(op_id); }

class CacheLookupOP : public Operator {
public:
    explicit CacheLookupOP(int op_id) : op_id_(op_id) {}
    void run(OpParams* params) override
    {
        FeatureCacheLookup(params->stream, (LegionUnifiedCache*)params->cache,
                           (LegionMemoryPool*)params->memorypool, op_id_, params->device_id);
        HIP_CALL(hipEventRecord(params->event, params->stream));
    }
private:
    int op_id_;
};
Operator* NewCacheLookupOP(int op_id) { return new CacheLookupOP(op_id); }

class SSDIOSubmitOP : public Operator {
public:
    explicit SSDIOSubmitOP(int op_id) : op_id_(op_id) {}
    void run(OpParams* params) override
    {
        IOSubmit(params->stream, (LegionFeatureStorage*)params->feature, (LegionMemoryPool*)params->memorypool,
                 op_id_, params->device_id);
        HIP_CALL(hipEventRecord(params->event, params->stream));
    }
private:
    int op_id_;
};
Operator* NewSSDIOSubmitOP(int op_id) { return new SSDIOSubmitOP(op_id); }

class SSDIOCompleteOP : public Operator {
public:
    explicit SSDIOCompleteOP(int op_id) : op_id_(op_id) {}
    void run(OpParams* params) override
    {
        MemoryPool* memorypool = (MemoryPool*)(params->memorypool);
        IOComplete(params->stream, (LegionUnifiedCache*)params->cache, (LegionMemoryPool*)memorypool,
                   params->device_id, memorypool->GetCurrentMode());
        HIP_CALL(hipEventRecord(params->event, params->stream));
    }
private:
    int op_id_;
};
Operator* NewSSDIOCompleteOP(int op_id) { return new SSDIOCompleteOP(op_id); }

// ---- kernel-level C entry points ----------------------------------------------------------
extern "C" void legion_gather_rows(legion_stream_t stream, const float* full_table,
                                   const float* const* cache_tables, const int32_t* node_map,
                                   int32_t node_capacity, int32_t float_feature_len, int32_t total_num_nodes,
                                   const int32_t* sampled_ids, int32_t* cache_index_out,
                                   const int32_t* range_devptr, float* dst, int32_t max_rows)
{
    lg::GatherParams g;
    g.replica = nullptr;
    g.replica_rows = 0;
    g.Kg = 1;
    g.member = 0;
    g.striped = true;               // the caller's node_map may address several tables: always decode (owner, row)
    g.stats = nullptr;
    g.full_table = full_table;
    g.cache_tables = cache_tables;
    g.local_table = nullptr;
    g.node_map = node_map;
    g.node_capacity = node_capacity;
    g.D = float_feature_len;
    g.skip_remote = false;
    g.grid_rows = 0;
    g.hybrid = false;
    g.hybrid_cpu_cap = g.hybrid_gpu_cap = 0;
    g.hybrid_cpu_cache = nullptr;
    g.total_num_nodes = total_num_nodes;
    g.max_rows = max_rows;
    lg::launch_gather_explicit(static_cast<hipStream_t>(stream), g, sampled_ids, cache_index_out, range_devptr, dst,
                               0x7FFFFFFF);
}

extern "C" void legion_draw_batch(legion_stream_t stream, const int32_t* idx, const int32_t* deg, int32_t* out,
                                  int32_t n)
{
    lg::launch_draw_batch(static_cast<hipStream_t>(stream), idx, deg, out, n);
}

// One whole mini-batch in the op order of GPURunner::RunOnce / RunPreSc (SS/engine/server.cu:285-332)
// without the IPC hand-off: what a Runner enqueues per batch, exposed for callers that own the
// buffers themselves (tests, bench.py, an in-process trainer).
static void enqueue_lanes(hipStream_t s, GraphStorage* graph, FeatureStorage* feature, UnifiedCache* cache,
                          const LanePtrs* d_lanes, int32_t n_lanes, MemoryPool* pool0, int32_t* iter_state,
                          int32_t batch_size, int32_t counter, int32_t dev_id, int32_t mode, bool is_presc,
                          const int32_t* fanout, int32_t hop_num, int32_t phase = LG_PHASE_ALL)
{
    if (cache == nullptr && !is_presc) {
        std::cout << "invalid cache ptr\n";     // serving needs the cache object (it owns the feature tiers)
        return;
    }
    if (cache && feature && cache->FeatureTable() == nullptr)
        cache->BindFeatureTable(feature->GetAllFloatFeature(), feature->TotalNodeNum());
    // inside a whole-batch enqueue every gather reads the {offset, count} snapshot its producer left in
    // hop_scratch[HS_RANGE + 2h] (not overwritten by later hops), so the gathers may also run as a
    // phase of their own after the whole sampler (LG_PHASE_GATHER, on another stream: pipeline.hip)
    const bool seeds_ride = hop_num >= 2;
    if (phase >= LG_PHASE_HEAD) {          // the two pieces of the weave arrangement (serve mode only)
        const int32_t last = hop_num - 1;
        // (every gather stays on the heavy stream, behind the last hop: the seeds' and earlier hops' gathers on the light stream
        // under the previous group's last gather were measured in round 4 -- no gain -- and removed)
        if (phase == LG_PHASE_HEAD) {
            do_batch_generate(s, feature, d_lanes, n_lanes, pool0, batch_size, counter, dev_id, mode, hop_num, iter_state);
            for (int32_t h = 0; h < last; h++)
                do_random_sample(s, graph, cache, d_lanes, n_lanes, pool0, fanout[h], dev_id, INTRABATCH_CON * (h + 1), false);
        } else {
            if (last >= 0) do_random_sample(s, graph, cache, d_lanes, n_lanes, pool0, fanout[last], dev_id, INTRABATCH_CON * (last + 1), false);
            lg::launch_end_of_batch(s, d_lanes, n_lanes, iter_state);
            if (phase == LG_PHASE_REST_SAMPLE) return;
            if (!seeds_ride) do_feature_lookup(s, cache, d_lanes, n_lanes, pool0, 1, dev_id, true);
            for (int32_t h = 0; h <= last; h++)
                do_feature_lookup(s, cache, d_lanes, n_lanes, pool0, INTRABATCH_CON * (h + 1) + 1, dev_id, true,
                                  (h == 0 && seeds_ride) ? 1 : -1);
        }
        return;
    }
    const bool sampler = phase != LG_PHASE_GATHER, gathers = phase != LG_PHASE_SAMPLE && !is_presc;
    if (sampler) do_batch_generate(s, feature, d_lanes, n_lanes, pool0, batch_size, counter, dev_id, mode, hop_num, iter_state);
    // The seeds' rows (op 1) are few and directly in front of hop 1's: when a later gather follows (so that
    // what FindFeat leaves in cache_search_buffer is the last op's either way), one launch fetches both.
    const bool seeds_ride_along = hop_num >= 2;
    if (gathers && !seeds_ride_along) do_feature_lookup(s, cache, d_lanes, n_lanes, pool0, 1, dev_id, true);
    for (int32_t h = 0; h < hop_num; h++) {
        const int32_t op = INTRABATCH_CON * (h + 1);
        if (sampler) do_random_sample(s, graph, cache, d_lanes, n_lanes, pool0, fanout[h], dev_id, op, is_presc);
        if (gathers) do_feature_lookup(s, cache, d_lanes, n_lanes, pool0, op + 1, dev_id, true,
                                       (h == 0 && seeds_ride_along) ? 1 : -1);
    }
    if (!sampler) return;
    if (is_presc && mode == TRAINMODE && cache != nullptr)     // CacheProfiling (one lane only in PreSC)
        cache->CacheProfiling(pool0->GetSampledIds(), pool0->GetAggSrcId(), pool0->GetAggDstId(), pool0->GetAggSrcOf(),
                              pool0->GetAggDstOf(), pool0->GetNodeCounter(), pool0->GetEdgeCounter(), s, dev_id);
    lg::launch_end_of_batch(s, d_lanes, n_lanes, iter_state);
}

extern "C" void legion_enqueue_batch(legion_stream_t strm_hdl, LegionGraphStorage* graph, LegionFeatureStorage* feature,
                                     LegionUnifiedCache* cache, LegionMemoryPool* memorypool, int32_t batch_size,
                                     int32_t counter, int32_t dev_id, int32_t mode, bool is_presc,
                                     const int32_t* fanout, int32_t hop_num)
{
    MemoryPool* mp = reinterpret_cast<MemoryPool*>(memorypool);
    if (!graph || !feature || !mp) { std::cout << "invalid storage ptr\n"; return; }
    enqueue_lanes(static_cast<hipStream_t>(strm_hdl), reinterpret_cast<GraphStorage*>(graph),
                  reinterpret_cast<FeatureStorage*>(feature), cache_of(cache), mp->DeviceLane(), 1, mp, mp->iter_state,
                  batch_size, counter, dev_id, mode, is_presc, fanout, hop_num);
}

// ---- lane groups: G independent mini-batches served by every launch ---------------------------
extern "C" LegionLaneGroup* legion_group_create(LegionMemoryPool** pools, int32_t n)
{
    LegionLaneGroup* g = new LegionLaneGroup();
    std::vector<LanePtrs> h;
    for (int32_t i = 0; i < n; i++) {
        MemoryPool* mp = reinterpret_cast<MemoryPool*>(pools[i]);
        g->pools.push_back(mp);
        h.push_back(mp->HostLane(mp->GetCurrentPipe()));
    }
    SetGPUDevice(g->pools[0]->dev_id);
    g->d_lanes = (LanePtrs*)d_alloc_space((int64_t)n * sizeof(LanePtrs));
    HIP_CALL(hipMemcpy(g->d_lanes, h.data(), h.size() * sizeof(LanePtrs), hipMemcpyHostToDevice));
    return g;
}

extern "C" void legion_group_set_iter_state(LegionLaneGroup* g, int32_t* iter_state_devptr) { if (g) g->iter_state = iter_state_devptr; }
// device address of lane `lane`'s descriptor (GPURunner: the source of a hand-over copy)
extern "C" const void* legion_group_lane_desc(LegionLaneGroup* g, int32_t lane)
{
    return (g && lane >= 0 && lane < (int32_t)g->pools.size()) ? (const void*)(g->d_lanes + lane) : nullptr;
}

extern "C" void legion_group_destroy(LegionLaneGroup* g)
{
    if (!g) return;
    d_free_space(g->d_lanes);
    delete g;
}

// Lane i of the group produces batch `counter0 + i` (serve mode).  One launch of every kernel covers
// all lanes (grid.y = lanes).
extern "C" void legion_enqueue_group_phase(legion_stream_t strm_hdl, LegionGraphStorage* graph, LegionFeatureStorage* feature,
                                           LegionUnifiedCache* cache, LegionLaneGroup* group, int32_t n_active,
                                           int32_t batch_size, int32_t counter0, int32_t dev_id, int32_t mode,
                                           const int32_t* fanout, int32_t hop_num, int32_t phase)
{
    if (!graph || !feature || !group || group->pools.empty()) { std::cout << "invalid storage ptr\n"; return; }
    if (n_active < 1 || n_active > (int32_t)group->pools.size()) n_active = (int32_t)group->pools.size();
    enqueue_lanes(static_cast<hipStream_t>(strm_hdl), reinterpret_cast<GraphStorage*>(graph),
                  reinterpret_cast<FeatureStorage*>(feature), cache_of(cache), group->d_lanes, n_active, group->pools[0],
                  group->iter_state, batch_size, counter0, dev_id, mode, false, fanout, hop_num, phase);
}

extern "C" void legion_enqueue_group_n(legion_stream_t strm_hdl, LegionGraphStorage* graph, LegionFeatureStorage* feature,
                                       LegionUnifiedCache* cache, LegionLaneGroup* group, int32_t n_active,
                                       int32_t batch_size, int32_t counter0, int32_t dev_id, int32_t mode,
                                       const int32_t* fanout, int32_t hop_num)
{
    legion_enqueue_group_phase(strm_hdl, graph, feature, cache, group, n_active, batch_size, counter0, dev_id, mode,
                               fanout, hop_num, LG_PHASE_ALL);
}

extern "C" void legion_enqueue_group(legion_stream_t strm_hdl, LegionGraphStorage* graph, LegionFeatureStorage* feature,
                                     LegionUnifiedCache* cache, LegionLaneGroup* group, int32_t batch_size,
                                     int32_t counter0, int32_t dev_id, int32_t mode, const int32_t* fanout,
                                     int32_t hop_num)
{
    legion_enqueue_group_n(strm_hdl, graph, feature, cache, group, 0, batch_size, counter0, dev_id, mode, fanout, hop_num);
}

// Diagnostics (bench.py's roofline.cold / roofline.warm_again, tools/): the gather of the group's LAST op once more, over the lanes as
// they stand -- same kernel instance, same grid, same ranges (the hop snapshot in hop_scratch) as inside a group's op list.
extern "C" void legion_enqueue_group_last_gather(legion_stream_t strm_hdl, LegionUnifiedCache* cache, LegionLaneGroup* group,
                                                 int32_t n_active, int32_t dev_id, int32_t hop_num)
{
    if (!cache || !group || group->pools.empty() || hop_num < 1) { std::cout << "invalid cache/group ptr\n"; return; }
    if (n_active < 1 || n_active > (int32_t)group->pools.size()) n_active = (int32_t)group->pools.size();
    do_feature_lookup(static_cast<hipStream_t>(strm_hdl), cache_of(cache), group->d_lanes, n_active, group->pools[0],
                      INTRABATCH_CON * hop_num + 1, dev_id, true, -1);
}

// ---- gather-op timing (HIP events recorded on the op's stream around the gather launch) -------
extern "C" void legion_pool_profile_begin(LegionMemoryPool* p_, int32_t max_ops)
{
    MemoryPool* mp = reinterpret_cast<MemoryPool*>(p_);
    if (!mp) return;
    SetGPUDevice(mp->dev_id);
    while ((int32_t)mp->prof_events.size() < 2 * max_ops) {
        hipEvent_t e;
        HIP_CALL(hipEventCreate(&e));
        mp->prof_events.push_back(e);
    }
    mp->prof_op.assign(max_ops, -1);
    mp->prof_used = 0;
    mp->prof_on = true;
}

// Stops recording; out_ms[i] / out_op[i] = elapsed time and op id of the i-th timed gather.
// The caller must have synchronised the stream.  Returns the number of timed ops.
extern "C" int32_t legion_pool_profile_end(LegionMemoryPool* p_, float* out_ms, int32_t* out_op, int32_t cap)
{
    MemoryPool* mp = reinterpret_cast<MemoryPool*>(p_);
    if (!mp) return 0;
    mp->prof_on = false;
    int32_t n = mp->prof_used < cap ? mp->prof_used : cap;
    for (int32_t i = 0; i < n; i++) {
        HIP_CALL(hipEventElapsedTime(&out_ms[i], mp->prof_events[2 * i], mp->prof_events[2 * i + 1]));
        out_op[i] = mp->prof_op[i];
    }
    return n;
}
