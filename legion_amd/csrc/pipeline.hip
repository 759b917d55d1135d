// pipeline.hip -- groups of mini-batches in flight on one GPU, each group replayed as one hipGraph.
//
// The reference's GPURunner (SS/engine/server.cu:302-332) issues ~16 launches per batch from the
// host, blocks on three 64-byte read-backs per hop, and keeps INTERBATCH_CON = 2 output slots so
// the trainer can consume batch i while batch i+1 is produced.  On MI355X one mini-batch is a chain
// of short, latency-bound kernels (3-8 us each at B = 1024) that cannot fill 256 CUs, ~16 eager
// launches cost more host time than the kernels take, and kernels of different HIP streams overlap
// only two at a time on this part (measured).  So:
//   * a GROUP of G independent mini-batches (lanes) is served by every launch: each kernel runs with
//     grid.y = G and lane g works on the buffers of pool g (LanePtrs).  The per-kernel latency floor
//     and launch cost are paid once per group, and the HBM-bound gather sees G times the rows;
//   * the whole op list of a group is captured once per (slot, mode) into a hipGraph; nothing in it
//     depends on host-side values: sizes (including the clamped last batch) are computed on the
//     device and the batch index lives in iter_state on the device, advanced by the last kernel;
//   * `slots` groups are in flight (default 2): while the consumer reads slot s, slot s+1 runs;
//   * weave (use_graph bit 4, what bench.py and the Runner use): a group is cut where its character changes -- the
//     HEAD (seeds + every hop but the last) of group k+1 runs on a low-priority stream under the heavy kernels of group k
//     (see submit).  The plain two-stream split (every sampler on one stream, every gather on another, with CU masks and
//     priorities) was measured against it in rounds 2-4 and removed in round 5: DESIGN_HISTORY.md 4.5.
#include "legion_core.h"

#include <unistd.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <thread>
#include <cstring>
#include <map>
#include <tuple>

struct LegionLaneGroup;
extern "C" void* d_alloc_scattered(int64_t num_bytes, int32_t chunk_mb);
extern "C" void* d_alloc_scattered_exportable(int64_t num_bytes, int32_t chunk_mb);
extern "C" int64_t lg_scattered_info(void* ptr, int32_t* n_chunks);
extern "C" int32_t lg_scattered_grant(void* ptr, const int32_t* logical_devs, int32_t n);
extern "C" int32_t lg_scattered_serve(void* ptr, const char* name);
extern "C" int32_t lg_scattered_exportable(void* ptr);
extern "C" void* lg_scattered_map_remote(const char* name, int32_t n_chunks, int64_t chunk_bytes);
extern "C" void lg_scattered_unmap_remote(void* base);
extern "C" void* lg_private_arena_begin(int64_t bytes, int32_t scatter_mb);
extern "C" void lg_private_arena_end();
extern "C" void lg_private_arena_free(void* handle);
extern "C" void lg_alloc_count_begin();
extern "C" int64_t lg_alloc_count_end();
extern "C" LegionLaneGroup* legion_group_create(LegionMemoryPool** pools, int32_t n);
extern "C" void legion_group_set_iter_state(LegionLaneGroup* g, int32_t* iter_state_devptr);
extern "C" void legion_group_destroy(LegionLaneGroup* g);
extern "C" void legion_enqueue_group_n(legion_stream_t strm_hdl, LegionGraphStorage* graph, LegionFeatureStorage* feature,
                                       LegionUnifiedCache* cache, LegionLaneGroup* group, int32_t n_active,
                                       int32_t batch_size, int32_t counter0, int32_t dev_id, int32_t mode,
                                       const int32_t* fanout, int32_t hop_num);
extern "C" void legion_enqueue_group_phase(legion_stream_t strm_hdl, LegionGraphStorage* graph, LegionFeatureStorage* feature,
                                           LegionUnifiedCache* cache, LegionLaneGroup* group, int32_t n_active,
                                           int32_t batch_size, int32_t counter0, int32_t dev_id, int32_t mode,
                                           const int32_t* fanout, int32_t hop_num, int32_t phase);
extern "C" void legion_pool_profile_begin(LegionMemoryPool* p_, int32_t max_ops);
extern "C" const void* legion_group_lane_desc(LegionLaneGroup* g, int32_t lane);

struct Slot {
    std::vector<MemoryPool*> pools;           // G lanes
    LegionLaneGroup* group = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;
    hipEvent_t sampled = nullptr;             // weave: the group's head has finished
    bool busy = false;
    std::map<std::tuple<int32_t, int32_t, int32_t, int32_t>, hipGraphExec_t> exec;   // key: (phase, mode, active lanes, batch_size)
    int32_t* d_iter = nullptr;                // device {next counter0, stride}
    int32_t* h_iter = nullptr;                // pinned staging
    int32_t next_iter = -1;                   // what d_iter[0] will hold once the slot is idle
    int32_t prof_pairs = 0;                   // timed gathers of the group in flight
};

// peer_gather = bulk (lg::BulkLists, kernels_gather.hip): per pipeline slot this GPU's request lists, and what it knows of
// the other members of its clique -- their lane arenas and their lists, as pointers this process can dereference
struct BulkPeer {
    char* arena = nullptr;
    bool mapped_chunks = false;                  // arena mapped from another process's file descriptors (unmapped at destroy)
    std::vector<lg::BulkLists> lists;            // [slot], pointers into the peer's memory
};
struct BulkState {
    int32_t Kg = 1, member = 0;
    int64_t cap = 0;
    std::vector<void*> alloc;                    // [slot] one exportable allocation: cnt | fidx | dst
    std::vector<lg::BulkLists> mine;             // [slot]
    std::vector<BulkPeer> peers;                 // [Kg]
    std::vector<hipStream_t> owner_streams;      // [Kg] in-process pull: a stream on each owner's device for THIS requester's pushes
    int32_t served_tag = -1;                     // >= 0: this arena's chunks are served on "legion_bulk_<pid>_<tag>"
};

struct LegionPipeline {
    GraphStorage* graph;
    FeatureStorage* feature;
    LegionUnifiedCache* cache_handle;
    int32_t dev_id, batch_size, hop_num, group_size, slots_n;
    std::vector<int32_t> fanout;
    std::vector<Slot> slots;
    bool use_graph;
    bool overlap = false;   // let kernels of different slots run concurrently (default: chained)
    bool weave = false;     // head of group k+1 on a second stream under the heavy kernels of group k (see submit)
    bool gathers = true;        // weave: false = the REST phase stops before the gathers (legion_pipeline_set_gathers)
    bool sample_only = false;   // only the sampler phase runs here; the owner gathers each lane itself (GPURunner: straight
                                // into a trainer-visible pipe slot)
    hipStream_t sample_stream = nullptr;      // weave: the light stream
    int32_t rr = 0;
    int32_t last_slot = -1;
    bool profiling = false;
    std::map<int32_t, double> prof_ms;        // op id -> summed elapsed ms of its gather launches
    std::map<int32_t, int64_t> prof_cnt;
    PoolArena arena;                          // use_graph bit 5: the lanes' trainer-visible arrays live in ONE exportable allocation
    BulkState* bulk = nullptr;
    void* priv_arena = nullptr;               // the block of shuffled chunks the lanes' private arrays were carved from (arena pipelines)
    bool arena_borrowed = false;              // the arena belongs to the caller (legion_pipeline_bulk_enable_shared)
    int64_t feature_rows = 0;
};

extern "C" LegionPipeline* legion_pipeline_create(LegionGraphStorage* graph, LegionFeatureStorage* feature,
                                                  LegionUnifiedCache* cache, int32_t dev_id, int32_t batch_size,
                                                  const int32_t* fanout, int32_t hop_num, int32_t group_size,
                                                  int32_t slots, int64_t feature_rows, int32_t use_graph)
{
    if (!graph || !feature || !cache) { printf("invalid storage ptr\n"); return nullptr; }
    LegionPipeline* p = new LegionPipeline();
    p->graph = reinterpret_cast<GraphStorage*>(graph);
    p->feature = reinterpret_cast<FeatureStorage*>(feature);
    p->cache_handle = cache;
    p->dev_id = dev_id;
    p->batch_size = batch_size;
    p->hop_num = hop_num;
    p->fanout.assign(fanout, fanout + hop_num);
    p->group_size = group_size < 1 ? 1 : group_size;
    p->slots_n = slots < 1 ? 1 : slots;
    p->use_graph = (use_graph & 1) != 0;
    p->overlap = (use_graph & 2) != 0;
    p->sample_only = (use_graph & 8) != 0;
    p->weave = (use_graph & 16) != 0;
    if (p->sample_only) p->weave = false;
    SetGPUDevice(dev_id);
    lg::tuning_refresh();
    if (p->weave) {
        // the light stream (heads of the next group) runs at LOW priority: its dozen small kernels have the whole rest of the
        // current group (1.6 ms for 0.2 ms of work) to finish, and at equal priority their workgroups take slots from the
        // dominant gather whenever both have some ready (measured: 5.11-5.12 -> 5.21-5.23 G edges/s at the headline)
        const int wp = lg::tuning().weave_priority;
        int lo = 0, hi = 0;                                  // hi is the numerically lowest = highest priority
        HIP_CALL(hipDeviceGetStreamPriorityRange(&lo, &hi));
        if (wp == 0) HIP_CALL(hipStreamCreateWithFlags(&p->sample_stream, hipStreamNonBlocking))
        else HIP_CALL(hipStreamCreateWithPriority(&p->sample_stream, hipStreamNonBlocking, wp > 0 ? hi : lo));
    }
    p->slots.resize(p->slots_n);
    p->feature_rows = feature_rows;
    if ((use_graph & 32) != 0) {       // one arena for every lane (peer_gather = bulk: the owners push rows into it)
        const int32_t D = p->feature->GetFloatFeatureLen();
        int64_t num_ids = batch_size, per = batch_size;
        for (int32_t h = 0; h < hop_num; h++) { per *= fanout[h]; num_ids += per; }
        p->arena.bytes = lg_pool_arena_bytes(batch_size, num_ids, feature_rows, D) * p->group_size * p->slots_n;
        // The arena is built from shuffled physical chunks (storage.hip d_alloc_scattered; LegionTuning.arena_scatter_mb = 0: one plain
        // allocation).  use_graph bit 6: it must be reachable from another process or GPU (peer_gather = bulk: owners push rows into
        // it) -- then its chunks are created exportable, other GPUs of this process are granted access when they are linked
        // (legion_pipeline_bulk_link) and other processes map them from file descriptors (legion_pipeline_bulk_export / _import).
        // (Round 4 kept such arenas plain: hipIpcGetMemHandle cannot export chunked memory.)
        const int32_t chunk_mb = lg::tuning().arena_scatter_mb;
        p->arena.base = (char*)(chunk_mb <= 0 ? d_alloc_space(p->arena.bytes)
                                              : ((use_graph & 64) != 0 ? d_alloc_scattered_exportable(p->arena.bytes, std::max(chunk_mb, 16))
                                                                       : d_alloc_scattered(p->arena.bytes, chunk_mb)));
        lg_set_pool_arena(&p->arena);
    }
    {   // what PreSC saw of the largest hop decides the small class's bucket count (8 or 16, storage.hip)
        int32_t last_hop[2] = {0, 0};
        if (p->cache_handle != nullptr)     // (the handle boxes the UnifiedCache as its first member, cache.hip)
            reinterpret_cast<UnifiedCache*>(p->cache_handle)->LastHopMax(dev_id, last_hop);
        lg_set_pool_claims_hint(last_hop[0], last_hop[1]);
    }
    // Arena pipelines (own arena or the caller's -- the server's) also carve their lanes' PRIVATE arrays from one block of shuffled
    // chunks (storage.hip): one lane is created with plain allocations first, only to add up what a lane asks for.
    if (lg_get_pool_arena() != nullptr && lg::tuning().arena_scatter_mb > 0) {
        PoolArena* trainer_arena = lg_get_pool_arena();
        lg_set_pool_arena(nullptr);
        lg_alloc_count_begin();
        LegionMemoryPool* probe = legion_pool_create(dev_id, p->feature->TotalNodeNum(), batch_size, fanout, hop_num, p->feature->GetFloatFeatureLen(), 1);
        if (feature_rows > 0) legion_pool_alloc_features(probe, feature_rows);
        const int64_t asked = lg_alloc_count_end();
        legion_pool_destroy(probe);
        lg_set_pool_arena(trainer_arena);
        int64_t num_ids = batch_size, per = batch_size;
        for (int32_t h = 0; h < hop_num; h++) { per *= fanout[h]; num_ids += per; }
        const int64_t visible = lg_pool_arena_bytes(batch_size, num_ids, feature_rows, p->feature->GetFloatFeatureLen());
        const int64_t lane = std::max<int64_t>(asked - visible, 0) + (64 << 10);
        p->priv_arena = lg_private_arena_begin(lane * p->group_size * p->slots_n + ((int64_t)8 << 20), lg::tuning().arena_scatter_mb);
    }
    for (Slot& sl : p->slots) {
        std::vector<LegionMemoryPool*> handles;
        for (int32_t g = 0; g < p->group_size; g++) {
            LegionMemoryPool* h = legion_pool_create(dev_id, p->feature->TotalNodeNum(), batch_size, fanout, hop_num,
                                                     p->feature->GetFloatFeatureLen(), 1);
            if (feature_rows > 0) legion_pool_alloc_features(h, feature_rows);
            handles.push_back(h);
            sl.pools.push_back(reinterpret_cast<MemoryPool*>(h));
        }
        sl.group = legion_group_create(handles.data(), p->group_size);
        sl.d_iter = (int32_t*)d_alloc_space(2 * sizeof(int32_t));
        HIP_CALL(hipMemset(sl.d_iter, 0, 2 * sizeof(int32_t)));
        HIP_CALL(hipHostMalloc((void**)&sl.h_iter, 2 * sizeof(int32_t), hipHostMallocDefault));
        // chained slots share one in-order stream (back-to-back graph launches, no event round trip);
        // overlapping slots get a stream each
        if (p->overlap || &sl == &p->slots[0])
            HIP_CALL(hipStreamCreateWithFlags(&sl.stream, hipStreamNonBlocking))
        else
            sl.stream = p->slots[0].stream;
        HIP_CALL(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
        HIP_CALL(hipEventCreateWithFlags(&sl.sampled, hipEventDisableTiming));
    }
    lg_set_pool_claims_hint(0, 0);
    lg_set_pool_arena(nullptr);
    if (p->priv_arena) lg_private_arena_end();
    return p;
}

static void slot_wait(LegionPipeline* p, Slot& sl)
{
    if (!sl.busy) return;
    HIP_CALL(hipEventSynchronize(sl.done));
    sl.busy = false;
    if (p->profiling) {                         // collect the HIP-event times of the finished group
        MemoryPool* mp = sl.pools[0];
        for (int32_t i = 0; i < sl.prof_pairs; i++) {
            float ms = 0.f;
            HIP_CALL(hipEventElapsedTime(&ms, mp->prof_events[2 * i], mp->prof_events[2 * i + 1]));
            p->prof_ms[mp->prof_op[i]] += ms;
            p->prof_cnt[mp->prof_op[i]] += 1;
        }
    }
    sl.prof_pairs = 0;
}

// The hipGraph of one phase of a group on a slot, captured on first use.  Nothing in it depends on host-side values: the
// iteration lives in sl.d_iter on the device (the group's iter_state must be set), sizes are computed by the kernels.
static hipGraphExec_t graph_of(LegionPipeline* p, Slot& sl, hipStream_t strm, int32_t phase, int32_t mode, int32_t n_active,
                               int32_t batch_size)
{
    // (a tuple, not packed bit fields: groups have up to 512 lanes, and (mode m, 256 + k lanes) must not meet (mode m + 1, k lanes))
    const auto key = std::make_tuple(phase, mode, n_active, batch_size);
    auto it = sl.exec.find(key);
    if (it == sl.exec.end()) {
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        HIP_CALL(hipStreamSynchronize(strm));
        HIP_CALL(hipStreamBeginCapture(strm, hipStreamCaptureModeThreadLocal));
        legion_enqueue_group_phase(strm, reinterpret_cast<LegionGraphStorage*>(p->graph), reinterpret_cast<LegionFeatureStorage*>(p->feature),
                                   p->cache_handle, sl.group, n_active, batch_size, 0, p->dev_id, mode, p->fanout.data(),
                                   p->hop_num, phase);
        HIP_CALL(hipStreamEndCapture(strm, &graph));
        HIP_CALL(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        HIP_CALL(hipGraphDestroy(graph));
        it = sl.exec.emplace(key, exec).first;
    }
    return it->second;
}

// A batch larger than the lanes were created for cannot be served: shrinking it silently would drop part of a validation
// or test set (their batches are sized from a raw 512, SS/engine/ipc_service.cu:91-115, whatever the training batch is)
// while the schedule's step counts stay the same.  The owner must size the pipeline for the largest batch of any mode
// (GPURunner::Initialize does); anything else is a bug and ends the process like every other error here.
static int32_t checked_batch_size(const LegionPipeline* p, int32_t batch_size)
{
    if (batch_size < 1) return p->batch_size;            // "the pipeline's own"
    if (batch_size > p->batch_size) {
        printf("legion_hip: batch size %d exceeds the %d the pipeline's lanes were created for\n", batch_size, p->batch_size);
        fflush(stdout);
        exit(EXIT_FAILURE);
    }
    return batch_size;
}

// Captures and instantiates, on EVERY slot, the graph(s) a later submit of (mode, n_active, batch_size) will replay, without
// launching anything.  A server calls this for every group shape of its schedule before it starts serving: stream capture
// and graph instantiation then never run beside another thread's HIP calls (GPURunner's poster polls events; a capture
// concurrent with hipEventQuery on the same stream crashed inside the runtime about once in thirty starts, ROCm 7.2).
extern "C" void legion_pipeline_prepare(LegionPipeline* p, int32_t mode, int32_t n_active, int32_t batch_size)
{
    if (!p || !p->use_graph) return;
    if (n_active < 1 || n_active > p->group_size) n_active = p->group_size;
    batch_size = checked_batch_size(p, batch_size);
    SetGPUDevice(p->dev_id);
    const int32_t first_phase = p->sample_only ? LG_PHASE_SAMPLE : LG_PHASE_ALL;
    for (Slot& sl : p->slots) {
        slot_wait(p, sl);
        legion_group_set_iter_state(sl.group, sl.d_iter);
        if (p->weave) {
            (void)graph_of(p, sl, p->sample_stream, LG_PHASE_HEAD, mode, n_active, batch_size);
            (void)graph_of(p, sl, sl.stream, p->gathers ? LG_PHASE_REST : LG_PHASE_REST_SAMPLE, mode, n_active, batch_size);
            continue;
        }
        (void)graph_of(p, sl, sl.stream, first_phase, mode, n_active, batch_size);
    }
}

// Enqueues the group of batches counter0 .. counter0 + G - 1 of `mode` on the next slot (round robin)
// and returns the slot index.  The slot's previous group must have been consumed: this call waits
// for its completion first.
extern "C" int32_t legion_pipeline_submit_n(LegionPipeline* p, int32_t counter0, int32_t mode, int32_t n_active);
extern "C" int32_t legion_pipeline_submit_ex(LegionPipeline* p, int32_t counter0, int32_t mode, int32_t n_active, int32_t batch_size);
extern "C" int32_t legion_pipeline_submit(LegionPipeline* p, int32_t counter0, int32_t mode)
{
    return legion_pipeline_submit_n(p, counter0, mode, p ? p->group_size : 0);
}
extern "C" int32_t legion_pipeline_submit_n(LegionPipeline* p, int32_t counter0, int32_t mode, int32_t n_active)
{
    return legion_pipeline_submit_ex(p, counter0, mode, n_active, p ? p->batch_size : 0);
}
extern "C" legion_stream_t legion_pipeline_stream(LegionPipeline* p) { return p ? (legion_stream_t)p->slots[0].stream : nullptr; }
// (GPURunner) the event recorded behind the last submitted group of a slot: another stream may wait for it
extern "C" void* legion_pipeline_slot_done_event(LegionPipeline* p, int32_t slot)
{
    return (p && slot >= 0 && slot < p->slots_n) ? (void*)p->slots[slot].done : nullptr;
}

// Same with only the first n_active lanes of the group working (the tail of a run whose length is
// not a multiple of the group size) and an explicit batch size (validation / test batches differ from
// training batches, SS/engine/ipc_service.cu:91-115; never larger than the pools were created for).
extern "C" int32_t legion_pipeline_submit_ex(LegionPipeline* p, int32_t counter0, int32_t mode, int32_t n_active, int32_t batch_size)
{
    if (!p) { printf("invalid pipeline ptr\n"); return -1; }
    if (n_active < 1 || n_active > p->group_size) n_active = p->group_size;
    batch_size = checked_batch_size(p, batch_size);
    SetGPUDevice(p->dev_id);
    const int32_t si = p->rr;
    p->rr = (p->rr + 1) % p->slots_n;
    Slot& sl = p->slots[si];
    lg::Range mark("group slot=%d first=%d lanes=%d mode=%d B=%d", si, counter0, n_active, mode, batch_size);
    slot_wait(p, sl);
    for (int32_t g = 0; g < p->group_size; g++) {
        sl.pools[g]->SetCurrentMode(mode);
        sl.pools[g]->SetIter(counter0 + g);
    }
    sl.pools[0]->prof_used = 0;
    // Chain the slots: this group starts on the GPU when the previously submitted one has finished.
    // The launch (and its host latency) still happens while that group runs, but kernels of different
    // groups never share the machine -- on this part two streams' kernels mostly take turns anyway,
    // and a gather that runs alone streams at ~73% of HBM peak instead of ~52%.
    // (chained slots share one stream, so the order is the stream's own)
    p->last_slot = si;
    LegionGraphStorage* gr = reinterpret_cast<LegionGraphStorage*>(p->graph);
    LegionFeatureStorage* f = reinterpret_cast<LegionFeatureStorage*>(p->feature);
    if (p->weave) {
        // Weave: the group is cut where its character changes.  HEAD (seeds + every hop but the last: a dozen small,
        // latency-bound kernels that cannot fill the chip) runs on the light stream Y; REST (the last hop -- scattered
        // atomics at ~2 TB/s of sector traffic -- and every gather) runs on the heavy stream X:
        //     X:  rest(k)                | rest(k+1)                 | ...
        //     Y:      head(k+1)          |      head(k+2)            |
        // The head of the next group hides under the current group's heavy kernels, which lose nothing measurable to it
        // (the dominant gather keeps 0.78 of the HBM peak); the two heavy kinds of traffic never share the machine, which
        // is what costs the plain two-stream split (sampler || gathers) a quarter of the gather's bandwidth.
        // Measured (RMAT-26, B = 1024, 256 lanes): 4.25-4.31 G edges/s against 4.06-4.15 G on one stream.  Running the
        // last hop's compaction kernels on Y beside the early gathers as well was measured too: no gain, gather at 0.75.
        hipStream_t X = sl.stream, Y = p->sample_stream;
        const bool eager = !p->use_graph || p->profiling;
        legion_group_set_iter_state(sl.group, eager ? nullptr : sl.d_iter);
        if (!eager && sl.next_iter != counter0) {
            sl.h_iter[0] = counter0;
            sl.h_iter[1] = p->group_size * p->slots_n;
            HIP_CALL(hipMemcpyAsync(sl.d_iter, sl.h_iter, 2 * sizeof(int32_t), hipMemcpyHostToDevice, Y));
        }
        auto run = [&](hipStream_t strm, int32_t phase) {
            if (eager)
                legion_enqueue_group_phase(strm, gr, f, p->cache_handle, sl.group, n_active, batch_size, counter0, p->dev_id, mode,
                                           p->fanout.data(), p->hop_num, phase);
            else
                HIP_CALL(hipGraphLaunch(graph_of(p, sl, strm, phase, mode, n_active, batch_size), strm));
        };
        run(Y, LG_PHASE_HEAD);
        HIP_CALL(hipEventRecord(sl.sampled, Y));
        HIP_CALL(hipStreamWaitEvent(X, sl.sampled, 0));
        run(X, p->gathers ? LG_PHASE_REST : LG_PHASE_REST_SAMPLE);
        if (eager) {
            sl.next_iter = -1;
            sl.prof_pairs = sl.pools[0]->prof_used;
        } else {
            sl.next_iter = n_active == p->group_size ? counter0 + p->group_size * p->slots_n : -1;
        }
        HIP_CALL(hipEventRecord(sl.done, X));
        sl.busy = true;
        return si;
    }
    hipStream_t s1 = sl.stream;
    const int32_t first_phase = p->sample_only ? LG_PHASE_SAMPLE : LG_PHASE_ALL;
    if (!p->use_graph || p->profiling) {            // HIP cannot time events recorded by graph nodes
        legion_group_set_iter_state(sl.group, nullptr);     // eager: iteration by value
        legion_enqueue_group_phase(s1, gr, f, p->cache_handle, sl.group, n_active, batch_size, counter0,
                                   p->dev_id, mode, p->fanout.data(), p->hop_num, first_phase);
        sl.next_iter = -1;
        sl.prof_pairs = sl.pools[0]->prof_used;
    } else {
        legion_group_set_iter_state(sl.group, sl.d_iter);
        if (sl.next_iter != counter0) {                     // (re)position the device-resident iteration
            sl.h_iter[0] = counter0;
            sl.h_iter[1] = p->group_size * p->slots_n;
            HIP_CALL(hipMemcpyAsync(sl.d_iter, sl.h_iter, 2 * sizeof(int32_t), hipMemcpyHostToDevice, s1));
        }
        HIP_CALL(hipGraphLaunch(graph_of(p, sl, s1, first_phase, mode, n_active, batch_size), s1));
        // what the last kernel leaves in d_iter[0] (a partial group breaks the regular stride)
        sl.next_iter = n_active == p->group_size ? counter0 + p->group_size * p->slots_n : -1;
    }
    HIP_CALL(hipEventRecord(sl.done, sl.stream));
    sl.busy = true;
    return si;
}

// Diagnostics: the last op's gather of the group sitting in `slot`, launched `repeats` more times over the lanes as they stand
// (a caller may have rewritten the lanes' ids in between: bench.py's cold-row figure), each launch between two HIP events on the
// slot's stream; ms_each[i] = its duration.  Returns the launches timed.
extern "C" void legion_enqueue_group_last_gather(legion_stream_t strm_hdl, LegionUnifiedCache* cache, LegionLaneGroup* group,
                                                 int32_t n_active, int32_t dev_id, int32_t hop_num);
extern "C" int32_t legion_pipeline_regather_last(LegionPipeline* p, int32_t slot, int32_t n_active, int32_t repeats, double* ms_each)
{
    if (!p || slot < 0 || slot >= p->slots_n || repeats < 1 || !ms_each) return 0;
    SetGPUDevice(p->dev_id);
    Slot& sl = p->slots[slot];
    slot_wait(p, sl);
    HIP_CALL(hipDeviceSynchronize());
    const bool was_on = sl.pools[0]->prof_on;
    sl.pools[0]->prof_on = false;                 // (this launch is timed here, not by the pool's op profile)
    hipEvent_t a, b;
    HIP_CALL(hipEventCreate(&a));
    HIP_CALL(hipEventCreate(&b));
    for (int32_t i = 0; i < repeats; i++) {
        HIP_CALL(hipEventRecord(a, sl.stream));
        legion_enqueue_group_last_gather(sl.stream, p->cache_handle, sl.group, n_active, p->dev_id, p->hop_num);
        HIP_CALL(hipEventRecord(b, sl.stream));
        HIP_CALL(hipEventSynchronize(b));
        float ms = 0.f;
        HIP_CALL(hipEventElapsedTime(&ms, a, b));
        ms_each[i] = ms;
    }
    HIP_CALL(hipEventDestroy(a));
    HIP_CALL(hipEventDestroy(b));
    sl.pools[0]->prof_on = was_on;
    return repeats;
}

extern "C" void legion_pipeline_wait(LegionPipeline* p, int32_t slot)
{
    if (!p) return;
    SetGPUDevice(p->dev_id);
    if (slot >= 0) { slot_wait(p, p->slots[slot % p->slots_n]); return; }
    for (Slot& sl : p->slots) slot_wait(p, sl);
}

// The same for a host thread that has nothing else to do until the group is complete (GPURunner handing batches over as views: a
// group completes every few ms): poll for at most spin_us microseconds, then SLEEP between polls (200 us naps: a completion is
// noticed at most that late, which delays a hand-over that has two more groups queued behind it, not the GPU).
// (hipEventSynchronize on an event created with hipEventBlockingSync was tried first: on this runtime the waiting thread and
// one of the runtime's own threads then burn 0.6 of a core each -- 1.25 cores per GPU at B = 8000 against 0.27 at B = 1024, where
// the thread mostly sleeps on the trainer's semaphore instead.)
extern "C" void legion_pipeline_wait_sleeping(LegionPipeline* p, int32_t slot, int32_t spin_us)
{
    if (!p) return;
    SetGPUDevice(p->dev_id);
    Slot& sl = p->slots[slot % p->slots_n];
    if (sl.busy) {
        const auto t0 = std::chrono::steady_clock::now();
        hipError_t q = hipEventQuery(sl.done);
        while (q == hipErrorNotReady && std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() < spin_us) q = hipEventQuery(sl.done);
        while (q == hipErrorNotReady) {
            std::this_thread::sleep_for(std::chrono::microseconds(200));
            q = hipEventQuery(sl.done);
        }
        (void)hipGetLastError();
        if (q != hipSuccess) HIP_CALL(q);
    }
    slot_wait(p, sl);       // (complete by now: returns at once, collects the profile)
}

extern "C" LegionMemoryPool* legion_pipeline_pool(LegionPipeline* p, int32_t slot, int32_t lane)
{
    if (!p) return nullptr;
    return reinterpret_cast<LegionMemoryPool*>(p->slots[slot % p->slots_n].pools[lane % p->group_size]);
}

// weave arrangement: whether the groups submitted (and the graphs prepared) from now on include their gathers.  GPURunner
// switches them off when the trainer end it serves gets its rows gathered batch by batch straight into a pipe slot.
extern "C" void legion_pipeline_set_gathers(LegionPipeline* p, int32_t on) { if (p) p->gathers = on != 0; }

extern "C" void legion_pipeline_destroy(LegionPipeline* p)
{
    if (!p) return;
    SetGPUDevice(p->dev_id);
    if (p->sample_stream) HIP_CALL(hipStreamSynchronize(p->sample_stream));
    for (Slot& sl : p->slots) {
        slot_wait(p, sl);
        HIP_CALL(hipStreamSynchronize(sl.stream));
        for (auto& kv : sl.exec) HIP_CALL(hipGraphExecDestroy(kv.second));
        legion_group_destroy(sl.group);
        d_free_space(sl.d_iter);
        HIP_CALL(hipHostFree(sl.h_iter));
        for (MemoryPool* mp : sl.pools) legion_pool_destroy(reinterpret_cast<LegionMemoryPool*>(mp));
        HIP_CALL(hipEventDestroy(sl.done));
        HIP_CALL(hipEventDestroy(sl.sampled));
        if (p->overlap || &sl == &p->slots[0]) HIP_CALL(hipStreamDestroy(sl.stream));
    }
    if (p->sample_stream) HIP_CALL(hipStreamDestroy(p->sample_stream));
    if (p->bulk) {
        for (size_t o = 0; o < p->bulk->owner_streams.size(); o++)
            if (p->bulk->owner_streams[o] != nullptr) {
                SetGPUDevice(p->dev_id / p->bulk->Kg * p->bulk->Kg + (int32_t)o);
                HIP_CALL(hipStreamDestroy(p->bulk->owner_streams[o]));
            }
        SetGPUDevice(p->dev_id);
        for (BulkPeer& peer : p->bulk->peers)
            if (peer.mapped_chunks && peer.arena != nullptr) lg_scattered_unmap_remote(peer.arena);
        for (void* a : p->bulk->alloc) d_free_space(a);
        delete p->bulk;
    }
    if (p->arena.base && !p->arena_borrowed) d_free_space(p->arena.base);
    if (p->priv_arena) lg_private_arena_free(p->priv_arena);
    delete p;
}

// ---- peer_gather = bulk: owner-bucketed transfer of the rows a striped gather needs from other members ----------------------
// (kernels_gather.hip bulk_bucket_kernel / bulk_push_kernel; LegionTuning.peer_gather.)  A group is produced in two host-visible
// phases, because the owners of the rows are other GPUs -- possibly other processes -- that must know the group's lists are final:
//   phase A (every member, its own group):  sampler -> bucket pass (per-owner lists) -> gather of everything that is NOT another
//            member's stripe (own stripe, replica, misses); stream synchronised on return
//   --- the caller's barrier over the clique (threads: any barrier; processes: e.g. torch.distributed) ---
//   phase B (every member, as an owner):    for every other member, push the rows it listed for this GPU into its lane arena;
//            stream synchronised on return
//   --- barrier --- : every lane of the slot holds its complete batch.
// Lookup results (cache_search_buffer, hit mask) and rows are those of the direct arrangement, bit for bit.
struct LegionBulkHandles {
    hipIpcMemHandle_t arena;          // arena_kind 0: a plain allocation
    hipIpcMemHandle_t lists[4];
    int64_t cap;
    int32_t slots, member;
    // arena_kind 1: the arena consists of arena_chunks chunks of arena_chunk_bytes, served as file descriptors on the abstract unix
    // socket "legion_bulk_<arena_pid>_<arena_tag>"
    int32_t arena_kind, arena_chunks, arena_pid, arena_tag;
    int64_t arena_chunk_bytes;
};
static_assert(sizeof(LegionBulkHandles) <= 512, "bulk handles travel in a 512-byte buffer");

extern "C" int32_t legion_pipeline_bulk_enable(LegionPipeline* p)
{
    if (!p || p->arena.base == nullptr) { printf("legion_hip: bulk transfers need a pipeline created with arena-backed lanes (use_graph bit 5)\n"); return 0; }
    if (p->slots_n > 4) { printf("legion_hip: at most 4 pipeline slots with bulk transfers\n"); return 0; }
    UnifiedCache* cache = reinterpret_cast<UnifiedCache*>(p->cache_handle);
    SetGPUDevice(p->dev_id);
    BulkState* b = new BulkState();
    b->Kg = cache->Kg_ > 0 ? cache->Kg_ : 1;
    b->member = p->dev_id % b->Kg;
    b->cap = (int64_t)p->group_size * std::max<int64_t>(p->feature_rows, 1);
    b->peers.resize(b->Kg);
    for (int32_t s = 0; s < p->slots_n; s++) {
        const int64_t bytes = 256 + b->Kg * b->cap * 12;
        char* a = (char*)d_alloc_space(bytes);
        HIP_CALL(hipMemset(a, 0, 256));
        lg::BulkLists l;
        l.Kg = b->Kg;
        l.cap = b->cap;
        l.cnt = (unsigned long long*)a;
        l.dst = (int64_t*)(a + 256);
        l.fidx = (int32_t*)(a + 256 + b->Kg * b->cap * 8);
        b->alloc.push_back(a);
        b->mine.push_back(l);
    }
    // the other members of the clique, as owners, write rows into THIS arena: a chunked arena must grant their devices access
    // (members that live in other processes map the chunks themselves: legion_pipeline_bulk_import)
    if (lg_scattered_info(p->arena.base, nullptr) > 0) {
        std::vector<int32_t> devs;
        const int32_t base = p->dev_id / b->Kg * b->Kg;
        for (int32_t o = 0; o < b->Kg; o++)
            if (lg_is_local(base + o)) devs.push_back(base + o);
        if (!lg_scattered_grant(p->arena.base, devs.data(), (int32_t)devs.size())) {
            for (void* a : b->alloc) d_free_space(a);
            delete b;
            return 0;
        }
    }
    p->bulk = b;
    return 1;
}

// the lanes were carved from an arena the CALLER owns (GPURunner's lane arena, lg_set_pool_arena before legion_pipeline_create):
// bulk transfers address it, the pipeline does not free it
extern "C" int32_t legion_pipeline_bulk_enable_shared(LegionPipeline* p, const PoolArena* arena)
{
    if (!p || !arena || arena->base == nullptr || p->arena.base != nullptr) return 0;
    p->arena = *arena;
    p->arena_borrowed = true;
    return legion_pipeline_bulk_enable(p);
}

static lg::BulkLists lists_at(char* a, int32_t Kg, int64_t cap)
{
    lg::BulkLists l;
    l.Kg = Kg;
    l.cap = cap;
    l.cnt = (unsigned long long*)a;
    l.dst = (int64_t*)(a + 256);
    l.fidx = (int32_t*)(a + 256 + Kg * cap * 8);
    return l;
}

// what the other members need of this GPU: IPC handles of its lane arena and of its lists (400 bytes)
extern "C" int32_t legion_pipeline_bulk_export(LegionPipeline* p, void* out_handles, int32_t out_bytes)
{
    if (!p || !p->bulk || !out_handles || out_bytes < (int32_t)sizeof(LegionBulkHandles)) return 0;
    SetGPUDevice(p->dev_id);
    LegionBulkHandles h;
    memset(&h, 0, sizeof(h));
    int32_t n_chunks = 0;
    const int64_t chunk_bytes = lg_scattered_info(p->arena.base, &n_chunks);
    if (chunk_bytes > 0) {              // an arena of shuffled chunks: other processes map it from file descriptors
        if (!lg_scattered_exportable(p->arena.base)) {      // (a pipeline created with arena = True, bit 5 without bit 6)
            printf("legion_hip: legion_pipeline_bulk_export: this pipeline's lane arena was not created exportable (use_graph bit 6 / arena = \"shared\"); "
                   "another process cannot map it\n");
            return 0;
        }
        static std::atomic<int32_t> next_tag{0};
        if (p->bulk->served_tag < 0) {
            p->bulk->served_tag = next_tag.fetch_add(1);
            char name[64];
            snprintf(name, sizeof(name), "legion_bulk_%d_%d", (int)getpid(), p->bulk->served_tag);
            if (!lg_scattered_serve(p->arena.base, name)) { printf("legion_hip: could not serve the lane arena's chunks on %s\n", name); return 0; }
        }
        h.arena_kind = 1;
        h.arena_chunks = n_chunks;
        h.arena_chunk_bytes = chunk_bytes;
        h.arena_pid = (int32_t)getpid();
        h.arena_tag = p->bulk->served_tag;
    } else {
        lg_ipc_export(&h.arena, p->arena.base, __FILE__, __LINE__);
    }
    for (int32_t s = 0; s < p->slots_n; s++) lg_ipc_export(&h.lists[s], p->bulk->alloc[s], __FILE__, __LINE__);
    h.cap = p->bulk->cap;
    h.slots = p->slots_n;
    h.member = p->bulk->member;
    memcpy(out_handles, &h, sizeof(h));
    return (int32_t)sizeof(h);
}

// a member that lives in ANOTHER process: open its handles
extern "C" int32_t legion_pipeline_bulk_import(LegionPipeline* p, const void* handles)
{
    if (!p || !p->bulk || !handles) return 0;
    LegionBulkHandles h;
    memcpy(&h, handles, sizeof(h));
    // (a member's lists have ITS capacity -- lanes x its own feature rows, which follow its own PreSC maximum)
    if (h.member < 0 || h.member >= p->bulk->Kg || h.member == p->bulk->member || h.slots != p->slots_n || h.cap <= 0) {
        printf("legion_hip: bulk handles of member %d do not fit this pipeline (slots %d/%d, cap %lld)\n", h.member, h.slots, p->slots_n,
               (long long)h.cap);
        return 0;
    }
    SetGPUDevice(p->dev_id);
    BulkPeer& peer = p->bulk->peers[h.member];
    void* a = nullptr;
    if (h.arena_kind == 1) {
        char name[64];
        snprintf(name, sizeof(name), "legion_bulk_%d_%d", h.arena_pid, h.arena_tag);
        a = lg_scattered_map_remote(name, h.arena_chunks, h.arena_chunk_bytes);
        if (a == nullptr) return 0;
        peer.mapped_chunks = true;
    } else {
        HIP_CALL(hipIpcOpenMemHandle(&a, h.arena, hipIpcMemLazyEnablePeerAccess));
    }
    peer.arena = (char*)a;
    peer.lists.clear();
    for (int32_t s = 0; s < h.slots; s++) {
        void* l = nullptr;
        HIP_CALL(hipIpcOpenMemHandle(&l, h.lists[s], hipIpcMemLazyEnablePeerAccess));
        peer.lists.push_back(lists_at((char*)l, p->bulk->Kg, h.cap));
    }
    return 1;
}

// a member that lives in THIS process (a thread per GPU, or logical GPUs of a test): take its pointers
extern "C" int32_t legion_pipeline_bulk_link(LegionPipeline* p, LegionPipeline* other)
{
    if (!p || !p->bulk || !other || !other->bulk || other->bulk->member == p->bulk->member || other->slots_n != p->slots_n) return 0;
    BulkPeer& peer = p->bulk->peers[other->bulk->member];
    peer.arena = other->arena.base;
    peer.lists = other->bulk->mine;
    // this GPU, as an owner, writes rows into the other member's arena: a chunked arena must say so (a plain one is reached through
    // the peer access SetGPUDevice enabled)
    const int32_t me = p->dev_id;
    if (lg_scattered_info(other->arena.base, nullptr) > 0 && !lg_scattered_grant(other->arena.base, &me, 1)) return 0;
    return 1;
}

// phase A; returns the slot
extern "C" int32_t legion_pipeline_bulk_phase_a(LegionPipeline* p, int32_t counter0, int32_t mode, int32_t n_active, int32_t batch_size)
{
    if (!p || !p->bulk) { printf("invalid pipeline ptr\n"); return -1; }
    if (n_active < 1 || n_active > p->group_size) n_active = p->group_size;
    batch_size = checked_batch_size(p, batch_size);
    SetGPUDevice(p->dev_id);
    const int32_t si = p->rr;
    p->rr = (p->rr + 1) % p->slots_n;
    Slot& sl = p->slots[si];
    lg::Range mark("bulk group A slot=%d first=%d lanes=%d", si, counter0, n_active);
    slot_wait(p, sl);
    for (int32_t g = 0; g < p->group_size; g++) {
        sl.pools[g]->SetCurrentMode(mode);
        sl.pools[g]->SetIter(counter0 + g);
    }
    hipStream_t X = sl.stream;
    LegionGraphStorage* gr = reinterpret_cast<LegionGraphStorage*>(p->graph);
    LegionFeatureStorage* f = reinterpret_cast<LegionFeatureStorage*>(p->feature);
    UnifiedCache* cache = reinterpret_cast<UnifiedCache*>(p->cache_handle);
    legion_group_set_iter_state(sl.group, nullptr);            // eager launches: the iteration by value
    legion_enqueue_group_phase(X, gr, f, p->cache_handle, sl.group, n_active, batch_size, counter0, p->dev_id, mode, p->fanout.data(),
                               p->hop_num, LG_PHASE_SAMPLE);
    const lg::BulkLists& mine = p->bulk->mine[si];
    HIP_CALL(hipMemsetAsync(mine.cnt, 0, 256, X));
    const LanePtrs* d_lanes = static_cast<const LanePtrs*>(legion_group_lane_desc(sl.group, 0));
    const int32_t max_rows = (int32_t)std::min<int64_t>(p->feature_rows, sl.pools[0]->num_ids);
    const int32_t last_op = INTRABATCH_CON * p->hop_num + 1;
    cache->BulkBucket(d_lanes, n_active, last_op, p->dev_id, X, max_rows, mine, p->arena.base);
    // the gathers in the op order of a whole-batch enqueue (operators.hip enqueue_lanes: the seeds ride along with hop 1 when a
    // later gather follows), each skipping the rows of other members' stripes
    const bool seeds_ride = p->hop_num >= 2;
    if (!seeds_ride) cache->FeatCacheLookup(d_lanes, n_active, 1, p->dev_id, X, max_rows, true, -1, p->hop_num == 0, true);
    for (int32_t h = 0; h < p->hop_num; h++)
        cache->FeatCacheLookup(d_lanes, n_active, INTRABATCH_CON * (h + 1) + 1, p->dev_id, X, max_rows, true, (h == 0 && seeds_ride) ? 1 : -1,
                               h + 1 == p->hop_num, /*skip_remote=*/true);
    HIP_CALL(hipStreamSynchronize(X));
    sl.next_iter = -1;
    p->last_slot = si;
    return si;
}

// phase B: this GPU as an owner, for the group every member has in `slot`
extern "C" void legion_pipeline_bulk_phase_b(LegionPipeline* p, int32_t slot)
{
    if (!p || !p->bulk || slot < 0 || slot >= p->slots_n) { printf("invalid pipeline ptr\n"); return; }
    SetGPUDevice(p->dev_id);
    lg::Range mark("bulk group B slot=%d", slot);
    Slot& sl = p->slots[slot];
    UnifiedCache* cache = reinterpret_cast<UnifiedCache*>(p->cache_handle);
    const int32_t me = p->bulk->member;
    for (int32_t r = 0; r < p->bulk->Kg; r++) {
        if (r == me) continue;
        const BulkPeer& peer = p->bulk->peers[r];
        if (peer.arena == nullptr || (int32_t)peer.lists.size() <= slot) {
            printf("legion_hip: member %d of the clique was never imported / linked\n", r);
            exit(EXIT_FAILURE);
        }
        const lg::BulkLists& l = peer.lists[slot];
        cache->BulkPush(p->dev_id, sl.stream, l.fidx + (int64_t)me * l.cap, l.dst + (int64_t)me * l.cap, l.cnt + me, l.cap, peer.arena);
    }
    HIP_CALL(hipStreamSynchronize(sl.stream));
    HIP_CALL(hipEventRecord(sl.done, sl.stream));
    sl.busy = true;
}

// One server process, a thread per GPU (the reference's deployment): the whole group in one call, no barrier.  Every member's
// stripe lives in this process, so the REQUESTER's thread itself launches the push kernels on the owners' devices (any thread
// may launch on any device; peer access is on: StorageManagement::EnableP2PAccess) as soon as its own lists are final, and waits
// for them.  Eager launches, synchronous: returns the slot with the group complete.
extern "C" int32_t legion_pipeline_submit_bulk_inproc(LegionPipeline* p, int32_t counter0, int32_t mode, int32_t n_active, int32_t batch_size)
{
    if (!p || !p->bulk) { printf("invalid pipeline ptr\n"); return -1; }
    const int32_t si = legion_pipeline_bulk_phase_a(p, counter0, mode, n_active, batch_size);
    if (si < 0) return si;
    BulkState* b = p->bulk;
    UnifiedCache* cache = reinterpret_cast<UnifiedCache*>(p->cache_handle);
    const int32_t base = p->dev_id / b->Kg * b->Kg, me = b->member;
    const lg::BulkLists& l = b->mine[si];
    if (b->owner_streams.empty()) b->owner_streams.assign(b->Kg, nullptr);
    for (int32_t o = 0; o < b->Kg; o++) {
        if (o == me) continue;
        SetGPUDevice(base + o);
        if (b->owner_streams[o] == nullptr) HIP_CALL(hipStreamCreateWithFlags(&b->owner_streams[o], hipStreamNonBlocking));
        cache->BulkPush(base + o, b->owner_streams[o], l.fidx + (int64_t)o * l.cap, l.dst + (int64_t)o * l.cap, l.cnt + o, l.cap, p->arena.base);
    }
    for (int32_t o = 0; o < b->Kg; o++) {
        if (o == me) continue;
        SetGPUDevice(base + o);
        HIP_CALL(hipStreamSynchronize(b->owner_streams[o]));
    }
    SetGPUDevice(p->dev_id);
    Slot& sl = p->slots[si];
    HIP_CALL(hipEventRecord(sl.done, sl.stream));
    sl.busy = true;
    return si;
}

// rows this GPU listed for the other members in `slot` (diagnostics: what crosses xGMI towards this GPU in phase B)
extern "C" int64_t legion_pipeline_bulk_listed(LegionPipeline* p, int32_t slot)
{
    if (!p || !p->bulk || slot < 0 || slot >= p->slots_n) return 0;
    SetGPUDevice(p->dev_id);
    unsigned long long c[8] = {0};
    HIP_CALL(hipMemcpy(c, p->bulk->mine[slot].cnt, sizeof(unsigned long long) * std::min(p->bulk->Kg, 8), hipMemcpyDeviceToHost));
    int64_t n = 0;
    for (int32_t i = 0; i < std::min(p->bulk->Kg, 8); i++) n += (int64_t)c[i];
    return n;
}

// Gather timing over the live pipeline: HIP events recorded on each slot's stream right before and
// after every gather launch.  hipEventElapsedTime rejects events recorded by graph nodes
// ('invalid resource handle' on ROCm 7.2), so groups submitted while profiling is on are launched
// eagerly (same kernels, same grouping).  read() returns, per gather op id, the summed elapsed ms and
// launch count of all groups that have been waited for since begin().
extern "C" void legion_pipeline_profile_begin(LegionPipeline* p)
{
    if (!p) return;
    SetGPUDevice(p->dev_id);
    for (Slot& sl : p->slots) {
        slot_wait(p, sl);
        HIP_CALL(hipStreamSynchronize(sl.stream));
        legion_pool_profile_begin(reinterpret_cast<LegionMemoryPool*>(sl.pools[0]), p->hop_num + 1);
    }
    p->prof_ms.clear();
    p->prof_cnt.clear();
    p->profiling = true;
}

extern "C" void legion_pipeline_profile_end(LegionPipeline* p)
{
    if (!p) return;
    for (Slot& sl : p->slots) {
        slot_wait(p, sl);
        sl.pools[0]->prof_on = false;
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
