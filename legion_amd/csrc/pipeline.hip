// pipeline.hip -- L mini-batches in flight per GPU, each replayed as one hipGraph.
//
// The reference's GPURunner (SS/engine/server.cu:302-332) issues ~16 launches per batch from the
// host, blocks on three 64-byte read-backs per hop, and keeps INTERBATCH_CON = 2 output slots so
// the trainer can consume batch i while batch i+1 is produced.  On MI355X one B=1024 batch is a
// chain of latency-bound kernels (3-7 us each) that cannot fill 256 CUs, and ~16 eager launches cost
// more host time than the kernels take.  So:
//   * a lane = {MemoryPool with its own outputs AND its own private scratch (position state,
//     compaction scratch), HIP stream, hipGraphExec}; lanes are independent, so L batches overlap
//     on the GPU (the reference's inter-batch pipe, generalised from "double-buffered outputs" to
//     "independent producers");
//   * the whole op list of a batch (legion_enqueue_batch) is captured once per (lane, mode) into a
//     hipGraph; nothing in it depends on host-side values: sizes are read from device counters
//     and the batch index lives in iter_state on the device (advanced by the last kernel);
//   * a batch whose size differs from the captured one (the clamped last batch of a set) and PreSC
//     batches run through the same entry point eagerly.
#include "legion_core.h"

#include <map>

extern "C" void legion_pool_profile_begin(LegionMemoryPool* p_, int32_t max_ops);
extern "C" void legion_enqueue_batch(legion_stream_t strm_hdl, LegionGraphStorage* graph, LegionFeatureStorage* feature,
                                     LegionUnifiedCache* cache, LegionMemoryPool* memorypool, int32_t batch_size,
                                     int32_t counter, int32_t dev_id, int32_t mode, bool is_presc,
                                     const int32_t* fanout, int32_t hop_num);

struct Lane {
    MemoryPool* pool = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;
    bool busy = false;
    std::map<int64_t, hipGraphExec_t> exec;   // key: mode * 2^32 + batch_size
    int32_t next_iter = -1;                   // value iter_state[0] will hold when the lane is idle
    int32_t* h_iter = nullptr;                // pinned {iter, stride} staging
    int32_t prof_pairs = 0;                   // timed gathers of the batch in flight
    std::map<int64_t, int32_t> exec_pairs;    // per captured graph
};

struct LegionPipeline {
    GraphStorage* graph;
    FeatureStorage* feature;
    UnifiedCache* cache;
    LegionUnifiedCache* cache_handle;
    int32_t dev_id, batch_size, hop_num, lanes_n;
    std::vector<int32_t> fanout;
    std::vector<Lane> lanes;
    bool use_graph;
    int32_t rr = 0;
    bool profiling = false;
    std::map<int32_t, double> prof_ms;        // op id -> summed elapsed ms of its gather launches
    std::map<int32_t, int64_t> prof_cnt;
};

static int32_t set_size_for(FeatureStorage* f, int32_t dev_id, int32_t mode)
{
    if (mode == TRAINMODE) return f->TrainingSetSize(dev_id);
    if (mode == VALIDMODE) return f->ValidationSetSize(dev_id);
    return f->TestingSetSize(dev_id);
}

extern "C" LegionPipeline* legion_pipeline_create(LegionGraphStorage* graph, LegionFeatureStorage* feature,
                                                  LegionUnifiedCache* cache, int32_t dev_id, int32_t batch_size,
                                                  const int32_t* fanout, int32_t hop_num, int32_t lanes,
                                                  int64_t feature_rows, int32_t use_graph)
{
    if (!graph || !feature || !cache) { printf("invalid storage ptr\n"); return nullptr; }
    LegionPipeline* p = new LegionPipeline();
    p->graph = reinterpret_cast<GraphStorage*>(graph);
    p->feature = reinterpret_cast<FeatureStorage*>(feature);
    p->cache_handle = cache;
    p->cache = reinterpret_cast<UnifiedCache*>(cache);
    p->dev_id = dev_id;
    p->batch_size = batch_size;
    p->hop_num = hop_num;
    p->fanout.assign(fanout, fanout + hop_num);
    p->lanes_n = lanes < 1 ? 1 : lanes;
    p->use_graph = use_graph != 0;
    SetGPUDevice(dev_id);
    p->lanes.resize(p->lanes_n);
    for (Lane& ln : p->lanes) {
        LegionMemoryPool* h = legion_pool_create(dev_id, p->feature->TotalNodeNum(), batch_size, fanout, hop_num,
                                                 p->feature->GetFloatFeatureLen(), 1);
        if (feature_rows > 0) legion_pool_alloc_features(h, feature_rows);
        ln.pool = reinterpret_cast<MemoryPool*>(h);
        ln.pool->iter_state = (int32_t*)d_alloc_space(2 * sizeof(int32_t));
        HIP_CALL(hipMemset(ln.pool->iter_state, 0, 2 * sizeof(int32_t)));
        HIP_CALL(hipHostMalloc((void**)&ln.h_iter, 2 * sizeof(int32_t), hipHostMallocDefault));
        HIP_CALL(hipStreamCreateWithFlags(&ln.stream, hipStreamNonBlocking));
        HIP_CALL(hipEventCreateWithFlags(&ln.done, hipEventDisableTiming));
    }
    return p;
}

static void lane_wait(LegionPipeline* p, Lane& ln)
{
    if (ln.busy) {
        HIP_CALL(hipEventSynchronize(ln.done));
        ln.busy = false;
        if (p->profiling) {                         // collect the HIP-event times of the finished batch
            for (int32_t i = 0; i < ln.prof_pairs; i++) {
                float ms = 0.f;
                HIP_CALL(hipEventElapsedTime(&ms, ln.pool->prof_events[2 * i], ln.pool->prof_events[2 * i + 1]));
                p->prof_ms[ln.pool->prof_op[i]] += ms;
                p->prof_cnt[ln.pool->prof_op[i]] += 1;
            }
        }
        ln.prof_pairs = 0;
    }
}

// Enqueues batch `counter` of `mode` on the next lane (round robin) and returns the lane index.
// The lane's previous batch must have been consumed: this call waits for its completion first.
extern "C" int32_t legion_pipeline_submit(LegionPipeline* p, int32_t counter, int32_t mode)
{
    if (!p) { printf("invalid pipeline ptr\n"); return -1; }
    SetGPUDevice(p->dev_id);
    const int32_t li = p->rr;
    p->rr = (p->rr + 1) % p->lanes_n;
    Lane& ln = p->lanes[li];
    lane_wait(p, ln);
    ln.pool->SetCurrentMode(mode);
    ln.pool->prof_used = 0;
    ln.pool->SetIter(counter);
    const int32_t total_cap = set_size_for(p->feature, p->dev_id, mode);
    const bool full = (int64_t)p->batch_size * (counter + 1) < total_cap;     // operator_impl.cu:159
    LegionGraphStorage* g = reinterpret_cast<LegionGraphStorage*>(p->graph);
    LegionFeatureStorage* f = reinterpret_cast<LegionFeatureStorage*>(p->feature);
    LegionMemoryPool* mp = reinterpret_cast<LegionMemoryPool*>(ln.pool);
    if (!p->use_graph || !full || p->profiling) {   // HIP cannot time events recorded by graph nodes
        int32_t* saved = ln.pool->iter_state;
        ln.pool->iter_state = nullptr;                  // eager: iteration by value
        legion_enqueue_batch(ln.stream, g, f, p->cache_handle, mp, p->batch_size, counter, p->dev_id, mode, false,
                             p->fanout.data(), p->hop_num);
        ln.pool->iter_state = saved;
        ln.next_iter = -1;
        ln.prof_pairs = ln.pool->prof_used;
    } else {
        if (ln.next_iter != counter) {                  // (re)position the device-resident iteration
            ln.h_iter[0] = counter;
            ln.h_iter[1] = p->lanes_n;
            HIP_CALL(hipMemcpyAsync(ln.pool->iter_state, ln.h_iter, 2 * sizeof(int32_t), hipMemcpyHostToDevice, ln.stream));
        }
        const int64_t key = ((int64_t)mode << 32) | (uint32_t)p->batch_size;
        auto it = ln.exec.find(key);
        if (it == ln.exec.end()) {
            hipGraph_t graph = nullptr;
            hipGraphExec_t exec = nullptr;
            HIP_CALL(hipStreamSynchronize(ln.stream));
            HIP_CALL(hipStreamBeginCapture(ln.stream, hipStreamCaptureModeThreadLocal));
            legion_enqueue_batch(ln.stream, g, f, p->cache_handle, mp, p->batch_size, counter, p->dev_id, mode, false,
                                 p->fanout.data(), p->hop_num);
            HIP_CALL(hipStreamEndCapture(ln.stream, &graph));
            HIP_CALL(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
            HIP_CALL(hipGraphDestroy(graph));
            it = ln.exec.emplace(key, exec).first;
            ln.exec_pairs[key] = ln.pool->prof_used;   // event-record nodes captured with the gathers
        }
        ln.prof_pairs = ln.exec_pairs[key];
        HIP_CALL(hipGraphLaunch(it->second, ln.stream));
        ln.next_iter = counter + p->lanes_n;            // what the last kernel leaves in iter_state[0]
    }
    HIP_CALL(hipEventRecord(ln.done, ln.stream));
    ln.busy = true;
    return li;
}

extern "C" void legion_pipeline_wait(LegionPipeline* p, int32_t lane)
{
    if (!p) return;
    SetGPUDevice(p->dev_id);
    if (lane >= 0) { lane_wait(p, p->lanes[lane % p->lanes_n]); return; }
    for (Lane& ln : p->lanes) lane_wait(p, ln);
}

extern "C" LegionMemoryPool* legion_pipeline_pool(LegionPipeline* p, int32_t lane)
{
    return p ? reinterpret_cast<LegionMemoryPool*>(p->lanes[lane % p->lanes_n].pool) : nullptr;
}

extern "C" legion_stream_t legion_pipeline_stream(LegionPipeline* p, int32_t lane)
{
    return p ? (legion_stream_t)p->lanes[lane % p->lanes_n].stream : nullptr;
}

extern "C" void legion_pipeline_destroy(LegionPipeline* p)
{
    if (!p) return;
    SetGPUDevice(p->dev_id);
    for (Lane& ln : p->lanes) {
        lane_wait(p, ln);
        HIP_CALL(hipStreamSynchronize(ln.stream));
        for (auto& kv : ln.exec) HIP_CALL(hipGraphExecDestroy(kv.second));
        d_free_space(ln.pool->iter_state);
        ln.pool->iter_state = nullptr;
        HIP_CALL(hipHostFree(ln.h_iter));
        legion_pool_destroy(reinterpret_cast<LegionMemoryPool*>(ln.pool));
        HIP_CALL(hipEventDestroy(ln.done));
        HIP_CALL(hipStreamDestroy(ln.stream));
    }
    delete p;
}

// Gather timing over the live pipeline: HIP events recorded on each lane's stream right before and
// after every gather launch.  hipEventElapsedTime rejects events recorded by graph nodes
// ('invalid resource handle' on ROCm 7.2), so batches submitted while profiling is on are launched
// eagerly (same kernels, same lanes).  read() returns, per gather op id, the summed elapsed ms and
// launch count of all batches that have been waited for since begin().
extern "C" void legion_pipeline_profile_begin(LegionPipeline* p)
{
    if (!p) return;
    SetGPUDevice(p->dev_id);
    for (Lane& ln : p->lanes) {
        lane_wait(p, ln);
        HIP_CALL(hipStreamSynchronize(ln.stream));
        legion_pool_profile_begin(reinterpret_cast<LegionMemoryPool*>(ln.pool), p->hop_num + 1);
    }
    p->prof_ms.clear();
    p->prof_cnt.clear();
    p->profiling = true;
}

extern "C" void legion_pipeline_profile_end(LegionPipeline* p)
{
    if (!p) return;
    for (Lane& ln : p->lanes) {
        lane_wait(p, ln);
        ln.pool->prof_on = false;
    }
    p->profiling = false;
}

extern "C" int32_t legion_pipeline_profile_read(LegionPipeline* p, int32_t* op_ids, double* ms_sums, int64_t* counts,
                                                int32_t cap)
{
    if (!p) return 0;
    int32_t n = 0;
    for (auto& kv : p->prof_ms) {
        if (n >= cap) break;
        op_ids[n] = kv.first;
        ms_sums[n] = kv.second;
        counts[n] = p->prof_cnt[kv.first];
        n++;
    }
    return n;
}
