// legion_core.h -- host-side object model of the MI355X sampling server.
//
// The class and method names mirror the reference's plugin/operator interface for the hot path
// (paths relative to the reference repo, SS = sampling_server/src):
//   Operator / OpParams            SS/engine/operator.h:4-28
//   Server / Runner / RunnerParams SS/engine/server.h:5-33
//   MemoryPool                     SS/engine/memorypool.cuh:20-221
//   GraphStorage                   SS/storage/graph_storage.cuh:7-24
//   FeatureStorage                 SS/storage/feature_storage.cuh:6-34
//   CacheController / UnifiedCache SS/cache/cache.cuh:10-177
//   IPCEnv                         SS/engine/ipc_service.h:6-35
// The implementations are new: device-resident counters (no host read-backs), first touches
// resolved bucket by bucket in LDS instead of bitmap + position map (no per-vertex state at all),
// direct-mapped id->slot tables instead of the vendored cuckoo hash, deterministic slot-ordered compaction.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/legion_hip.h"

#define INTERBATCH_CON LEGION_INTERBATCH_CON
#define INTRABATCH_CON LEGION_INTRABATCH_CON
#define MAX_DEVICE LEGION_MAX_DEVICE
#define MEMORY_USAGE LEGION_MEMORY_USAGE
#define TRAINMODE LEGION_TRAINMODE
#define VALIDMODE LEGION_VALIDMODE
#define TESTMODE LEGION_TESTMODE
#define CACHEMISS_FLAG LEGION_CACHEMISS_FLAG

// Error convention of the reference (cudaCheckError, SS/engine/operator_impl.cu:16-24): print and exit.
#define hipCheckError()                                                                       \
    {                                                                                         \
        hipError_t e_ = hipGetLastError();                                                    \
        if (e_ != hipSuccess) {                                                               \
            printf("HIP failure %s:%d: '%s'\n", __FILE__, __LINE__, hipGetErrorString(e_));   \
            exit(EXIT_FAILURE);                                                               \
        }                                                                                     \
    }
#define HIP_CALL(expr)                                                                        \
    {                                                                                         \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) {                                                               \
            printf("HIP failure %s:%d: '%s'\n", __FILE__, __LINE__, hipGetErrorString(e_));   \
            exit(EXIT_FAILURE);                                                               \
        }                                                                                     \
    }

// ---- first touches without per-vertex state (replaces accessed_map + position_map) ----------
// The reference marks first touches with atomicOr on an N-bit map it memsets per batch and keeps positions in an N-entry map it
// clears node by node (operator_impl.cu:151,244-272,542-548).  Here a hop issues no memory-side atomic per claim and keeps
// nothing per vertex: a hop's claims (vertex, slot) are written, grouped by hash bucket, into one list per bucket of the lane
// (ranks by LDS atomics, one reservation per bucket and tile); a second kernel gives every (lane, bucket) a workgroup that builds
// an open-addressing table of the bucket's vertices IN LDS -- the batch's known vertices of that bucket (the seeds from
// sampled_ids, the nodes earlier hops added from the bucket's known list, which list_known_kernel keeps) with their positions,
// then the claims with atomicMin on (vertex, pending | slot) -- and marks every claim that is not the lowest slot of a new
// vertex.  Nothing survives the hop, nothing to clear, nothing that scales with N.  A bucket whose vertices do not fit the
// table is handled in several passes over sub-buckets, so the result never depends on the hash.
// Buckets per lane follow the pool's largest hop, so that a bucket sees a few thousand claims: 8 (or 16, dense graphs) up to 2^19
// slots per lane (B = 1024-class batches), 64 up to 2^22 (B = 8000 with [25,10]), 256 beyond (B = 8000 with [15,10,5] has 6 M,
// with [25,10,10] 20 M; larger hops run more passes per bucket).  The kernels are instantiated for the four classes.
// (Rounds 1-4 also had a uint32[N] array per lane and a per-lane open-addressing table in memory, claimed with one atomicMin per
// pick: slower by 5-14 % wherever the LDS form applied, removed in round 5 when it applied everywhere -- DESIGN_HISTORY.md.)
#define LG_LDS_BITS_SMALL 3
#define LG_LDS_BITS_SMALL16 4                    // the same class with 16 buckets: dense graphs, see lg_set_pool_claims_hint
#ifndef LG_LDS_BITS_MEDIUM
#define LG_LDS_BITS_MEDIUM 6
#endif
#ifndef LG_LDS_BITS_LARGE
#define LG_LDS_BITS_LARGE 8
#endif
#ifndef LG_DEDUP_CLAIMS
#define LG_DEDUP_CLAIMS 5                       // claims a thread of a de-duplication workgroup keeps in registers (a bucket of at most LG_DEDUP_CLAIMS x 1024 is "resident")
#endif
#ifndef LG_DEDUP_CLAIMS_MID
#define LG_DEDUP_CLAIMS_MID 10                  // ... 10 where PreSC saw buckets of 5-10 k claims (B = 8000 on the less repetitive graphs: uk-union size, RMAT-28): 64 KB
#endif                                          // table, two workgroups per CU as with 5
#ifndef LG_DEDUP_CLAIMS_BIG
#define LG_DEDUP_CLAIMS_BIG 20                  // ... 20 beyond (B = 8000 [15,10,5] on RMAT-26: 15 k per bucket)
#endif
#ifndef LG_DEDUP_BIG_TABLE_BITS
#define LG_DEDUP_BIG_TABLE_BITS 14               // ... and the log2 words of its LDS table (14: 128 KB)
#endif
#define LG_LDS_SLOTS_SMALL (1 << 19)
#define LG_LDS_SLOTS_MEDIUM (1 << 22)
// fewest super tiles (1024 slots) a partition tile of the 64- / 256-bucket classes may have (the launch picks up to
// LG_PLACE_MAX_K, kernels_sample.hip); sizes run_off
static inline int32_t lg_lds_k_min(int32_t bucket_bits) { return bucket_bits == LG_LDS_BITS_LARGE ? 4 : 1; }
#ifndef LG_LDS_TABLE_BITS
#define LG_LDS_TABLE_BITS 13
#endif
#define LG_LDS_TABLE (1 << LG_LDS_TABLE_BITS)   // 64-bit words of LDS per (lane, bucket) workgroup
#ifndef LG_LDS_FILL_16THS
#define LG_LDS_FILL_16THS 14                    // a pass may fill its table up to this many sixteenths (bound: known + claims of the pass)
#endif
__host__ __device__ inline uint32_t lg_tab_hash(int32_t id)
{
    uint32_t x = (uint32_t)id;
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// per-hop scratch written by the scan kernel, read by scatter / localise (device int32[16])
enum HopScratch {
    HS_FRONTIER_IS_SEEDS = 0,
    HS_FRONTIER_OFF = 1,   // offset of the frontier inside agg_src_ids / agg_src_off
    HS_FRONTIER_LEN = 2,
    HS_NODE_BASE = 3,
    HS_EDGE_BASE = 4,
    HS_N_NEW = 5,
    HS_N_EDGE = 6,
    HS_SLOTS = 7,
    HS_CTICKET = 24,       // compact_kernel: next super tile to hand out / workgroups that finished (both zero between hops)
    HS_CDONE = 25,
    HS_ERROR = 29,         // sticky error bits of the lane (LG_ERR_*), also mirrored to the pool's host-visible flag
    HS_RANGE = 10,         // [HS_RANGE + 2h], [+1]: {offset, count} of the new nodes of op 3h, kept for its gather
    HS_WORDS = 32
};

// error bits a kernel can raise for its lane (MemoryPool::ErrorBits / legion_pool_error)
#define LG_ERR_TABLE_FULL 1       // a de-duplication bucket did not fit its LDS table even in 2^14 sub-bucket passes (not a hash problem: cannot happen)
#define LG_ERR_FEATURE_ROWS 2     // the batch has more rows than the feature buffer: the gather stopped at its end
#define LG_ERR_CHAIN 4            // compact_kernel gave up waiting for an earlier tile's status word / a winner's position (cannot happen)

// Device code: a pointer that was loaded from memory (LanePtrs, pointer tables, LDS) is "generic" to
// the compiler, which then emits flat_* instructions; those count on lgkmcnt as well as vmcnt, so every
// LDS wait also waits for the memory operations in flight.  Kernels cast such pointers to the global
// address space once (every buffer of this library is HBM, peer HBM or mapped host memory).
#define LG_G __attribute__((address_space(1)))
#define LG_GPTR(T, p) ((LG_G T*)(p))

// phases of a whole-batch enqueue (legion_enqueue_group_phase)
#define LG_PHASE_ALL 0      // reference op order: gather right after the op that produced its rows
#define LG_PHASE_SAMPLE 1   // BatchGenerate + every RandomSample + IOComplete
#define LG_PHASE_GATHER 2   // every FeatureCacheLookup, from the per-op range snapshots
// "weave" arrangement (pipeline.hip): the group cut where its character changes
#define LG_PHASE_HEAD 3     // BatchGenerate + every hop but the last, complete: small, latency-bound kernels
#define LG_PHASE_REST 4     // the last hop (sample .. localise) + IOComplete + every gather, in that order
#define LG_PHASE_REST_SAMPLE 5   // LG_PHASE_REST without the gathers (GPURunner serving a trainer end that gets its rows gathered
                                 // batch by batch straight into a pipe slot)

// Feature-cache slot of a sampled neighbour, carried from the sampler to the gather (see "column slots", GraphStorage):
// a value >= 0 or CACHEMISS_FLAG is what node_map[id] holds; LG_FS_UNKNOWN means "not carried: look it up"
#define LG_FS_UNKNOWN (-3)
// slot_dst[slot] of a hop: -1 no edge; v >= 0 the sampled neighbour; v < -1 the neighbour -2 - v, marked by the de-duplication as
// NOT the first touch of a new vertex (an involution: the same expression marks and unmarks; every int32 vertex id fits)
#define LG_SLOT_LOSER(v) (-2 - (v))
#ifndef LG_CLAIM_CNT_STRIDE
#define LG_CLAIM_CNT_STRIDE 32        // ints between the claim-list counts of two buckets: a line each (the 8 / 16 reservations of a super tile go to different lines)
#endif
#define LG_CLAIM_CHUNK_BITS 9
#define LG_CLAIM_CHUNK (1 << LG_CLAIM_CHUNK_BITS)

#define LG_TILE 256            // compaction tile == threads per workgroup in the sampler kernels
#define LG_SLOTS_PER_LANE 4    // independent slots each lane keeps in flight
#define LG_SUPER (LG_TILE * LG_SLOTS_PER_LANE)   // slots one workgroup owns per iteration

// Per-vertex row header: where the adjacency of v lives (slot of the CSR pointer tables: P = the
// full CSR, d < P = GPU d's cached CSR), its first edge and its degree.  One 16-byte read resolves
// what the reference does with two hash finds + two indptr reads (cache.cu:217-225,
// operator_impl.cu:224-230).
struct alignas(16) RowHdr {
    int64_t start;
    int32_t deg;
    int32_t slot;
};

struct BuildInfo {  // SS/include/buildinfo.h (only the fields of the in-memory path)
    int32_t partition_count = 0;
    std::vector<int32_t> training_set_num, validation_set_num, testing_set_num;
    std::vector<std::vector<int32_t>> training_set_ids, training_labels;
    std::vector<std::vector<int32_t>> validation_set_ids, validation_labels;
    std::vector<std::vector<int32_t>> testing_set_ids, testing_labels;
    int32_t total_num_nodes = 0;
    int32_t float_feature_len = 0;
    float* host_float_feature = nullptr;   // device-dereferenceable (HBM or mapped pinned)
    int64_t* csr_node_index = nullptr;     // device-dereferenceable
    int32_t* csr_dst_node_ids = nullptr;
    int64_t total_edge_num = 0;
    int32_t epoch = 0;
    int32_t raw_batch_size = 0;
};

// Everything a kernel needs to know about ONE in-flight mini-batch (a lane): the buffers of its
// MemoryPool.  Arrays of these live in device memory; kernels are launched with grid.y = number of
// lanes and each workgroup works on lanes[blockIdx.y], so one launch serves a whole group of
// independent mini-batches (the per-kernel latency floor is paid once per group, not per batch).
struct LanePtrs {
    int32_t* sampled_ids;
    int32_t* labels;
    int32_t* agg_src_ids;
    int32_t* agg_dst_ids;
    int32_t* agg_src_off;
    int32_t* agg_dst_off;
    char* tmp_part_ind;
    // first touches (no per-vertex state, see above): the hop's claims, one list per hash bucket
    unsigned long long* claim_pairs;   // (vertex << 32 | slot): [buckets][claim_cap], interleaved by chunk (below)
    int32_t* run_off;                  // 256-bucket class: [partition tiles][buckets] x {first place in the bucket's claim list, count} (sample_kernel -> place_kernel)
    int32_t lds_buckets;               // 8, 16, 64 or 256
    // one list of claims per bucket: the sampling kernel reserves places for a super tile's (8/16 buckets) or a partition tile's
    // (64/256 buckets) claims of a bucket with one atomic on claim_cnt[bucket]; a count beyond claim_cap says the list is incomplete
    // and the bucket's workgroup reads the hop's slots instead (kernels_sample.hip)
    // The lists are interleaved in chunks of LG_CLAIM_CHUNK entries -- entry k of bucket b sits at lg_claim_at(b, k, buckets) --
    // so that what a hop really uses (a fraction of the capacity) is one dense prefix of the array, as few pages as the segment form
    // touches (measured: with one contiguous region per bucket, 16 x oversized, both kernels lost ~15 us per group to translation misses)
    int32_t* claim_cnt;                // [buckets], zero between hops
    int32_t claim_cap;
    int32_t ids_cap;                   // capacity of sampled_ids (what may be read before the live counters are known)
    // ... and the batch's vertices that later hops must recognise (every node but the seeds and the last hop's), one list
    // per bucket: scatter appends (vertex << 32 | position), the next hop's workgroup of that bucket reads only its list
    unsigned long long* known_pairs;   // [buckets][known_cap]
    int32_t* known_cnt;                // [buckets] entries appended (may exceed known_cap: then the list is not used)
    int32_t known_cap;
    int32_t* err_flag;                 // mapped pinned host word: kernels OR LG_ERR_* bits into it
    int32_t* counter_mirror;           // mapped pinned host int32[32] or null: the end-of-batch kernel leaves the batch's
                                       // node_counter / edge_counter there (GPURunner's lanes: no device read-back per hand-over)
    const void* deliver;               // lg::DeliverParams* (device) or null: the gather of this lane also hands its batch
                                       // over to that trainer-visible pipe slot (GPURunner's hand-over descriptors)
    int32_t* node_counter;
    int32_t* edge_counter;
    int32_t* slot_dst;
    int32_t* slot_pos;                 // [max_slots] -1, or for a loser: the vertex's position / -2 - (the winning slot); winners publish their position here
    int32_t* slot_fs;                  // [max_slots] feature-cache slot of the slot's sampled neighbour, or LG_FS_UNKNOWN (column slots)
    int32_t* node_slot;                // [num_ids] the same per node of the batch, by position in sampled_ids: what the gather reads
                                       // instead of node_map[id] (one 128-byte line per row for 4 bytes)
    unsigned long long* tile_state;    // [super tiles] look-back status words of compact_kernel (zero between hops)
    int32_t* hop_scratch;
    RowHdr* fh_edge;
    int32_t* cache_search_buffer;
    float* float_features;
    int32_t feature_rows;
    int32_t max_slots;
};

// ---------------------------------------------------------------------------------------------
class MemoryPool {
public:
    explicit MemoryPool(int32_t pipeline_depth)
        : pipeline_depth_(pipeline_depth), float_features_(pipeline_depth, nullptr),
          labels_(pipeline_depth, nullptr), node_counter_(pipeline_depth, nullptr),
          edge_counter_(pipeline_depth, nullptr), sampled_ids_(pipeline_depth, nullptr),
          agg_src_off_(pipeline_depth, nullptr), agg_dst_off_(pipeline_depth, nullptr) {}

    int32_t GetIter() const { return iter_; }
    int32_t GetCurrentMode() const { return mode_; }
    int32_t GetCurrentPipe() const { return current_pipe_; }
    int32_t PipelineDepth() const { return pipeline_depth_; }
    float* GetFloatFeatures() const { return float_features_[current_pipe_]; }
    int32_t* GetCacheSearchBuffer() const { return cache_search_buffer_; }
    int32_t* GetLabels() const { return labels_[current_pipe_]; }
    int32_t* GetPositionMap() const { return nullptr; }      // (the reference's accessor, memorypool.cuh:120-135: no such array here)
    int32_t* GetNodeCounter() const { return node_counter_[current_pipe_]; }
    int32_t* GetEdgeCounter() const { return edge_counter_[current_pipe_]; }
    int32_t* GetSampledIds() const { return sampled_ids_[current_pipe_]; }
    int32_t* GetAggSrcId() const { return agg_src_ids_; }
    int32_t* GetAggDstId() const { return agg_dst_ids_; }
    int32_t* GetAggSrcOf() const { return agg_src_off_[current_pipe_]; }
    int32_t* GetAggDstOf() const { return agg_dst_off_[current_pipe_]; }
    char* GetTmpPartIdx() const { return tmp_part_ind_; }
    int32_t* GetTmpPartOff() const { return tmp_part_off_; }

    void SetFloatFeatures(float* p, int32_t pipe) { float_features_[pipe] = p; lanes_dirty_ = true; }
    void SetCacheSearchBuffer(int32_t* p) { cache_search_buffer_ = p; lanes_dirty_ = true; }
    void SetLabels(int32_t* p, int32_t pipe) { labels_[pipe] = p; lanes_dirty_ = true; }
    void SetNodeCounter(int32_t* p, int32_t pipe) { node_counter_[pipe] = p; lanes_dirty_ = true; }
    void SetEdgeCounter(int32_t* p, int32_t pipe) { edge_counter_[pipe] = p; lanes_dirty_ = true; }
    void SetSampledIds(int32_t* p, int32_t pipe) { sampled_ids_[pipe] = p; lanes_dirty_ = true; }
    void SetAggSrcId(int32_t* p) { agg_src_ids_ = p; lanes_dirty_ = true; }
    void SetAggDstId(int32_t* p) { agg_dst_ids_ = p; lanes_dirty_ = true; }
    void SetAggSrcOf(int32_t* p, int32_t pipe) { agg_src_off_[pipe] = p; lanes_dirty_ = true; }
    void SetAggDstOf(int32_t* p, int32_t pipe) { agg_dst_off_[pipe] = p; lanes_dirty_ = true; }
    void SetTmpPartIdx(char* p) { tmp_part_ind_ = p; lanes_dirty_ = true; }
    void SetTmpPartOff(int32_t* p) { tmp_part_off_ = p; }
    void SetCurrentPipe(int32_t pipe) { current_pipe_ = pipe; }
    LanePtrs HostLane(int32_t pipe) const;          // the pool's buffers of one pipe slot
    const LanePtrs* DeviceLane();                    // device copy of HostLane(current pipe)
    void InvalidateDeviceLanes() { lanes_dirty_ = true; }
    void SetCurrentMode(int32_t mode) { mode_ = mode; }
    void SetIter(int32_t iter) { iter_ = iter; }

    // new in this build: sampler scratch (all device memory, private to the server)
    int32_t* slot_dst = nullptr;       // [max_slots] sampled neighbour per slot
    int32_t* slot_pos = nullptr;       // [max_slots] see LanePtrs
    int32_t* slot_fs = nullptr;        // [max_slots] / [num_ids]: feature-cache slots carried from the sampler to the gather
    int32_t* node_slot = nullptr;
    unsigned long long* tile_state = nullptr;   // [max_tiles / 4 + 1] (LanePtrs)
    RowHdr* fh_edge = nullptr;         // [num_ids] frontier row headers written by scatter
    int32_t* hop_scratch = nullptr;    // [HS_WORDS]
    unsigned long long* claim_pairs = nullptr; // see LanePtrs
    int32_t* run_off = nullptr;
    int32_t* claim_cnt = nullptr;
    int32_t claim_cap = 0;
    int32_t lds_bucket_bits = 0;
    int64_t last_hop_claims_hint = 0;  // PreSC's maximum of the last hop's edges (0: unknown), see lg_pool_alloc_private
    unsigned long long* known_pairs = nullptr;
    int32_t* known_cnt = nullptr;
    int32_t known_cap = 0;
    int32_t* err_host = nullptr;       // host-visible error word (mapped pinned), err_dev = its device address
    int32_t* err_dev = nullptr;
    int32_t ErrorBits() const { return err_host ? *(volatile int32_t*)err_host : 0; }
    // lanes of a GPURunner group (lg_set_pool_arena): the trainer-visible arrays live in the runner's arena (not freed
    // here) and the counters are mirrored to host-visible memory by the end-of-batch kernel
    bool arena_backed = false;
    int32_t* counter_mirror_dev = nullptr;
    int32_t* counter_mirror_host = nullptr;
    int32_t num_ids = 0;
    int32_t max_slots = 0;             // largest hop = B * f1 * ... * fH
    std::vector<int64_t> max_new;      // [h] upper bound of new nodes of op 3h (h = 0: the seeds)
    int32_t total_num_nodes = 0;
    int32_t batch_size = 0;
    int32_t float_feature_len = 0;
    int64_t feature_rows = 0;
    int64_t grid_rows_hint = 0;        // > 0: rows a batch typically has (the Runner's pipe-slot pool holds the worst case: launches are sized for the usual one)
    int32_t dev_id = 0;
    bool owns_buffers = false;

    // graph replay: {next iteration, stride} on the device; when set, BatchGenerate reads the
    // iteration there and IOComplete advances it (no per-replay kernel-argument update)
    int32_t* iter_state = nullptr;

    // optional per-op timing with HIP events on the op's own stream (bench.py roofline leg)
    std::vector<hipEvent_t> prof_events;   // pairs: [2*i] before, [2*i+1] after
    std::vector<int32_t> prof_op;          // op id of each pair
    int32_t prof_used = 0;
    bool prof_on = false;

    void Finalize();

private:
    int32_t iter_ = 0;
    int32_t mode_ = 0;
    int32_t* cache_search_buffer_ = nullptr;
    int32_t* agg_src_ids_ = nullptr;
    int32_t* agg_dst_ids_ = nullptr;
    char* tmp_part_ind_ = nullptr;
    int32_t* tmp_part_off_ = nullptr;
    int32_t pipeline_depth_;
    int32_t current_pipe_ = 0;
    LanePtrs* d_lanes_ = nullptr;      // [pipeline_depth] device copies
    bool lanes_dirty_ = true;
    int64_t uploaded_rows_ = -1;
    std::vector<float*> float_features_;
    std::vector<int32_t*> labels_;
    std::vector<int32_t*> node_counter_;
    std::vector<int32_t*> edge_counter_;
    std::vector<int32_t*> sampled_ids_;
    std::vector<int32_t*> agg_src_off_;
    std::vector<int32_t*> agg_dst_off_;
};

// ---------------------------------------------------------------------------------------------
class GraphStorage {
public:
    virtual ~GraphStorage() = default;
    virtual void Build(BuildInfo* info) = 0;
    virtual void GraphCache(int32_t* QT, int32_t Ki, int32_t Kg, int32_t capacity) = 0;
    virtual void Finalize() = 0;
    virtual int32_t GetPartitionCount() const = 0;
    virtual int64_t** GetCSRNodeIndex(int32_t part_id) const = 0;
    virtual int32_t** GetCSRNodeMatrix(int32_t part_id) const = 0;
    virtual int64_t* GetCSRNodeIndexCPU() const = 0;
    virtual int32_t* GetCSRNodeMatrixCPU() const = 0;
    virtual int32_t NodeNum() const = 0;
    virtual int64_t EdgeNum() const = 0;
    virtual const RowHdr* GetRowHeaders(int32_t part_id) const = 0;   // new: [N] per GPU
    // "Column slots" (new): a copy of the full column array in which every entry is the pair {neighbour id, feature-cache
    // slot of that neighbour = node_map[id]} (8 bytes).  The sampler's scattered 4-byte pick costs a whole 64-byte sector
    // either way; read as 8 bytes it brings the neighbour's cache slot along for free, and the gather no longer fetches a
    // 128-byte line of node_map per row (measured: 10 % of the hop-2 gather's traffic at D = 128, 19 % at D = 64).  Built per
    // GPU after FillUp from that GPU's node_map (which is what makes it clique-specific), only when the column array is
    // device memory and 8 bytes per edge are affordable (LegionTuning.col_slots); slots of cached-topology CSRs are not
    // paired: their picks carry LG_FS_UNKNOWN and the gather looks those rows up as before.
    virtual int32_t** GetCSRXMatrix(int32_t part_id) const = 0;       // device table [P+1] of pair arrays (null entries), or null
    virtual const int32_t* GetColumnSlotsFull(int32_t part_id) const = 0;   // that GPU's pair copy of the full column array, or null
    // The pairs are only as good as the node_map they were built from: they carry the stamp (cache uid, fill generation) of
    // that fill, and the sampler is given them only by a cache whose CURRENT fill has the same stamp (ColumnSlotsStamp) -- a
    // re-fill with other capacities, or another cache object over the same graph, samples from the plain column array and
    // the gather looks the slots up (a stale carried slot would address a wrong or out-of-range cache row).  Launch groups
    // captured into hipGraphs hold these pointers: destroy pipelines before re-filling a cache.
    virtual void BuildColumnSlots(int32_t dev, const int32_t* node_map, uint64_t stamp = 0) = 0;
    virtual void DropColumnSlots(int32_t dev) = 0;
    virtual uint64_t ColumnSlotsStamp(int32_t dev) const = 0;     // stamp of the fill the pairs belong to; 0: none built
    // GraphCache in two steps, so that a clique spread over processes can exchange the stripes in
    // between: build the cached CSR of every LOCAL member, then link every known member's CSR into the
    // local members' pointer tables and row headers.  SetPeerCSR registers a member owned by another
    // process (pointers opened from its IPC handles).
    virtual void GraphCacheBuildLocal(int32_t* QT, int32_t Ki, int32_t Kg, int32_t capacity) = 0;
    virtual void GraphCacheLink(int32_t* QT, int32_t Ki, int32_t Kg, int32_t capacity) = 0;
    virtual void SetPeerCSR(int32_t dev, int64_t* csr_node_index, int32_t* csr_dst_node_ids) = 0;
    virtual int64_t* CachedCSRIndex(int32_t dev) const = 0;
    virtual int32_t* CachedCSRDst(int32_t dev) const = 0;
};
extern "C" GraphStorage* NewCompleteGraphStorage();

class FeatureStorage {
public:
    virtual ~FeatureStorage() = default;
    virtual void Build(BuildInfo* info, int in_memory_mode) = 0;
    virtual void Finalize() = 0;
    virtual int32_t* GetTrainingSetIds(int32_t part_id) const = 0;
    virtual int32_t* GetValidationSetIds(int32_t part_id) const = 0;
    virtual int32_t* GetTestingSetIds(int32_t part_id) const = 0;
    virtual int32_t* GetTrainingLabels(int32_t part_id) const = 0;
    virtual int32_t* GetValidationLabels(int32_t part_id) const = 0;
    virtual int32_t* GetTestingLabels(int32_t part_id) const = 0;
    virtual int32_t TrainingSetSize(int32_t part_id) const = 0;
    virtual int32_t ValidationSetSize(int32_t part_id) const = 0;
    virtual int32_t TestingSetSize(int32_t part_id) const = 0;
    virtual int32_t TotalNodeNum() const = 0;
    virtual float* GetAllFloatFeature() const = 0;
    virtual int32_t GetFloatFeatureLen() const = 0;
    // SSD tier: unreleased in the reference (feature_storage.cu:146-154 are TODO stubs); kept as no-ops
    virtual void IOSubmit(int32_t*, int32_t*, int32_t*, float*, int32_t, int32_t, hipStream_t) {}
    virtual void IOComplete() {}
    // plain-buffer set-up used by the C API
    virtual void SetIds(int32_t dev_id, int32_t mode, const int32_t* host_ids,
                        const int32_t* host_labels, int32_t count) = 0;
};
extern "C" FeatureStorage* NewCompleteFeatureStorage();

// ---------------------------------------------------------------------------------------------
class CacheController {  // SS/cache/cache.cuh:10-62, the cache-policy plug-in point
public:
    virtual ~CacheController() = default;
    virtual void Initialize(int32_t dev_id, int32_t total_num_nodes) = 0;
    virtual void Finalize() = 0;
    virtual void FindFeat(int32_t* sampled_ids, int32_t* cache_offset, int32_t* node_counter,
                          int32_t op_id, void* stream) = 0;
    virtual void FindTopo(int32_t* input_ids, char* partition_index, int32_t* partition_offset,
                          int32_t batch_size, int32_t op_id, void* strm_hdl, int32_t device_id) = 0;
    virtual void CacheProfiling(int32_t* sampled_ids, int32_t* agg_src_id, int32_t* agg_dst_id,
                                int32_t* agg_src_off, int32_t* agg_dst_off, int32_t* node_counter,
                                int32_t* edge_counter, bool is_presc, void* stream) = 0;
    virtual void InitializeMap(int node_capacity, int edge_capacity) = 0;
    virtual void Insert(int32_t* QT, int32_t* QF, int32_t cache_expand, int32_t Kg) = 0;
    virtual void HybridInsert(int32_t* QF, int32_t cpu_cache_capacity, int32_t gpu_cache_capacity) = 0;
    virtual void AccessCount(int32_t* d_key, int32_t num_keys, void* stream) = 0;
    virtual unsigned long long int* GetNodeAccessedMap() = 0;
    virtual unsigned long long int* GetEdgeAccessedMap() = 0;
    virtual unsigned long long int* GetTopoTransactions() = 0;   // device counter, see sample_kernel
    virtual int32_t MaxIdNum() = 0;
    // direct-mapped id->value tables (the bcht::find contract), device pointers, may be null
    virtual const int32_t* NodeMap() const = 0;
    virtual const char* EdgeIndexMap() const = 0;
    virtual const int32_t* EdgeOffsetMap() const = 0;
};
CacheController* NewPreSCCacheController(int32_t train_step, int32_t device_count);

namespace lg { struct BulkLists; struct GatherParams; }

class UnifiedCache {
public:
    void Initialize(int64_t cache_memory, int32_t float_feature_len, int32_t train_step,
                    int32_t device_count, int32_t cpu_cache_capacity, int32_t gpu_cache_capacity);
    void InitializeCacheController(int32_t dev_id, int32_t total_num_nodes);
    void Finalize(int32_t dev_id);
    void FindFeat(int32_t* sampled_ids, int32_t* cache_offset, int32_t* node_counter, int32_t op_id,
                  void* stream, int32_t dev_id);
    void FindTopo(int32_t* input_ids, char* partition_index, int32_t* partition_offset,
                  int32_t batch_size, int32_t op_id, void* strm_hdl, int32_t dev_id);
    void CacheProfiling(int32_t* sampled_ids, int32_t* agg_src_id, int32_t* agg_dst_id,
                        int32_t* agg_src_off, int32_t* agg_dst_off, int32_t* node_counter,
                        int32_t* edge_counter, void* stream, int32_t dev_id);
    void CandidateSelection(int cache_agg_mode, FeatureStorage* feature, GraphStorage* graph);
    void CostModel(int cache_agg_mode, FeatureStorage* feature, GraphStorage* graph,
                   std::vector<uint64_t>& counters, int32_t train_step);
    void FillUp(int cache_agg_mode, FeatureStorage* feature, GraphStorage* graph);
    void FillUpLocal(FeatureStorage* feature, GraphStorage* graph);   // maps + the local members' stripes
    void FillUpLink(FeatureStorage* feature, GraphStorage* graph);    // pointer tables over every known member
    // The hybrid CPU-cache / GPU-cache tier (SS/cache/cache.cu:614-670; the reference's server.cu:112 keeps the call commented
    // out): per GPU, its OWN hotness order; the gpu_cache_capacity hottest rows in an HBM cache, the next cpu_cache_capacity
    // in a mapped pinned host cache, the rest misses.  Replaces CandidateSelection + CostModel + FillUp.  miss_from_table:
    // a miss row is read from the FeatureStorage table (this build's stand-in for the unreleased SSD reader); false = the
    // reference's kernel to the letter: the gather leaves a miss row unwritten.
    void HybridInit(FeatureStorage* feature, GraphStorage* graph, bool miss_from_table = true);
    void SetHybridCapacity(int32_t cpu_cache_capacity, int32_t gpu_cache_capacity)
    {
        cpu_cache_capacity_ = cpu_cache_capacity;
        gpu_cache_capacity_ = gpu_cache_capacity;
    }
    int32_t CPUCapacity() const { return cpu_cache_capacity_; }     // cache.cu:676-682
    int32_t GPUCapacity() const { return gpu_cache_capacity_; }
    bool IsHybrid() const { return hybrid_; }
    const float* HybridCPUCache(int32_t dev_id) const { return dev_id < (int32_t)cpu_cache_dev_.size() ? cpu_cache_dev_[dev_id] : nullptr; }
    void SetPeerFeatureCache(int32_t dev, float* ptr) { float_feature_cache_[dev] = ptr; }
    float* FeatureCachePtr(int32_t dev) const { return float_feature_cache_[dev]; }
    void SetPeerMaxIds(const int32_t* v, int32_t n) { peer_max_ids_.assign(v, v + n); }
    // Hot-row replica (new; MI355X has 288 GB per GPU): besides its stripe of the clique's cache every member keeps a
    // private copy of the clique's hottest rows, as many as `bytes` hold.  The lookup result (hit mask, global slot) is
    // unchanged -- only where a hit row is READ changes: ranks below the replica size come from local HBM instead of a
    // peer over xGMI.  Set before FillUp.
    void SetReplicaMemory(int64_t bytes) { replica_bytes_ = bytes; }
    int32_t ReplicaRows(int32_t dev_id) const { return replica_rows_.empty() ? 0 : replica_rows_[dev_id]; }
    int32_t FloatFeatureLen() const { return float_feature_len_; }
    bool gather_stats_on_ = true;        // GatherStats() arms the counters; SetGatherStats pauses them (a device word: graphs follow)
    void SetGatherStats(bool on);
    unsigned long long* GatherStats(int32_t dev_id);   // device {stripe rows, replica rows, peer-stripe rows}, allocated on first use
    int32_t MaxIdNum(int32_t dev_id);
    void LastHopMax(int32_t dev_id, int32_t out[2]);   // PreSC maxima {edges of the last hop, nodes before it}; {0, 0} before PreSC
    unsigned long long int* GetEdgeAccessedMap(int32_t dev_id);
    // the gather over a group of lanes (the reference's per-array arguments live in LanePtrs)
    // first_op_id < op_id: one launch also covers the new-node ranges of the earlier ops first_op_id, +3, ...
    void FeatCacheLookup(const LanePtrs* d_lanes, int32_t n_lanes, int32_t op_id, int32_t dev_id,
                         hipStream_t strm_hdl, int32_t max_rows, bool use_snapshot, int32_t first_op_id = -1, bool last_op = true,
                         bool skip_remote = false, int32_t grid_rows = 0);
    // peer_gather = bulk (lg::BulkLists): the requester's bucket pass over every row of its group's batches, and the owner's push
    void BulkBucket(const LanePtrs* d_lanes, int32_t n_lanes, int32_t op_id, int32_t dev_id, hipStream_t s, int32_t max_rows,
                    const lg::BulkLists& lists, const char* arena_base);
    lg::GatherParams GatherParamsOf(int32_t dev_id, int32_t op_id, int32_t max_rows, bool use_snapshot, int32_t first_op_id, bool last_op);
    void BulkPush(int32_t owner_dev, hipStream_t s, const int32_t* fidx, const int64_t* dst, const unsigned long long* cnt,
                  int64_t cap, char* peer_arena);

    // new / exposed for the C API and the fused kernels
    void SetCapacity(int32_t node_capacity, int32_t edge_capacity);
    void BindFeatureTable(float* table, int32_t total_num_nodes) { cpu_float_features_ = table; total_num_nodes_ = total_num_nodes; }
    float* FeatureTable() const { return cpu_float_features_; }
    int32_t NodeCapacity(int32_t dev_id) const;
    int32_t EdgeCapacity(int32_t dev_id) const;
    CacheController* Controller(int32_t dev_id) const { return cache_controller_[dev_id]; }
    bool IsPresc() const { return is_presc_; }
    // (uid of this cache object, generation of its latest FillUpLocal): what column slots built from its node_map carry
    uint64_t FillStamp() const { return (uid_ << 24) | (fill_generation_ & 0xFFFFFFu); }
    bool world_reduced = false;   // hotness already all-reduced across processes (RCCL)
    // how the last CandidateSelection summed the counters of each clique: 0 nothing to sum / taken as they are, 1 the leader
    // loop over peer pointers, 2 RCCL all-reduce; and the milliseconds the RCCL calls took
    std::vector<int32_t> hotness_reduce_path_;
    double hotness_reduce_ms_ = 0;
    int32_t total_num_nodes_ = 0;
    int32_t Kc_ = 1, Kg_ = 1;
    int cache_agg_mode_ = 0;
    std::vector<int32_t*> QF_, QT_;
    std::vector<unsigned long long int*> AF_, AT_;
    float** Global_Float_Feature_Cache(int32_t dev_id) const { return d_float_feature_cache_ptr_[dev_id]; }

private:
    int32_t device_count_ = 0;
    std::vector<CacheController*> cache_controller_;
    std::vector<int32_t> node_capacity_, edge_capacity_;
    int32_t cpu_cache_capacity_ = 0, gpu_cache_capacity_ = 0;
    int64_t cache_memory_ = 0;
    std::vector<float*> float_feature_cache_;
    std::vector<float**> d_float_feature_cache_ptr_;
    int32_t float_feature_len_ = 0;
    uint64_t uid_ = 0, fill_generation_ = 0;
    float* cpu_float_features_ = nullptr;
    bool is_presc_ = true;
    std::vector<int32_t> peer_max_ids_;   // MaxIdNum of every clique member when they live in other processes
    int64_t replica_bytes_ = 0;
    std::vector<float*> replica_;         // [device] hottest rows of its clique in rank order, or null
    std::vector<int32_t> replica_rows_;
    std::vector<unsigned long long*> gather_stats_;
    // hybrid tier: one CPU cache per GPU (the reference allocates a single one on device 0 although every GPU orders its rows
    // by its own counters, cache.cu:616,630-641 -- "single gpu version", :681)
    bool hybrid_ = false, hybrid_miss_from_table_ = true;
    std::vector<float*> cpu_cache_dev_, cpu_cache_host_;   // device-side / host-side address of each GPU's mapped pinned cache
    float* hybrid_table_ = nullptr;                        // the FeatureStorage table misses are served from
};

// ---------------------------------------------------------------------------------------------
class IPCEnv {  // SS/engine/ipc_service.h:6-33
public:
    virtual ~IPCEnv() = default;
    virtual void Coordinate(BuildInfo* info) = 0;
    virtual int32_t GetMaxStep() = 0;
    virtual void InitializeSamplesBuffer(int32_t batch_size, int32_t num_ids, int32_t feature_dim,
                                         int32_t device_id, int32_t pipeline_depth) = 0;
    virtual void InitializeFeaturesBuffer(int32_t batch_size, int32_t num_ids, int32_t feature_dim,
                                          int32_t device_id, int32_t pipeline_depth) = 0;
    virtual int32_t GetRawBatchsize() = 0;
    virtual int32_t GetLocalBatchId(int32_t global_batch_id) = 0;
    virtual int32_t GetCurrentBatchsize(int32_t dev_id, int32_t current_mode) = 0;
    virtual int32_t GetCurrentMode(int32_t global_batch_id) = 0;
    virtual int32_t* GetIds(int32_t dev_id, int32_t current_pipe) = 0;
    virtual float* GetFloatFeatures(int32_t dev_id, int32_t current_pipe) = 0;
    virtual int32_t* GetLabels(int32_t dev_id, int32_t current_pipe) = 0;
    virtual int32_t* GetAggSrc(int32_t dev_id, int32_t current_pipe) = 0;
    virtual int32_t* GetAggDst(int32_t dev_id, int32_t current_pipe) = 0;
    virtual int32_t* GetNodeCounter(int32_t dev_id, int32_t current_pipe) = 0;
    virtual int32_t* GetEdgeCounter(int32_t dev_id, int32_t current_pipe) = 0;
    virtual void IPCPost(int32_t dev_id, int32_t current_pipe) = 0;
    virtual void IPCWait(int32_t dev_id, int32_t current_pipe) = 0;
    // new in this build (see ipc_env.hip): host-visible per-slot counter mirror written by the GPU, non-blocking wait
    virtual int32_t* GetCounterMirror(int32_t dev_id, int32_t current_pipe) = 0;
    virtual void PublishMirror() = 0;
    virtual bool IPCTryWait(int32_t dev_id, int32_t current_pipe) = 0;
    // direct-view hand-over (round 4, ipc_env.hip shmExt): the lane arena's IPC handle, whether the trainer end of dev_id
    // takes views of it, and the per-slot description of the batch being posted
    virtual int32_t* HostCounterMirror(int32_t dev_id, int32_t current_pipe) = 0;
    virtual bool PublishArena(int32_t dev_id, void* base, int64_t bytes) = 0;
    virtual bool TrainerTakesViews(int32_t dev_id) = 0;
    virtual void SetView(int32_t dev_id, int32_t pipe, const int64_t* off5, const int32_t* counters32) = 0;
    virtual void AbortServing() = 0;
    virtual void Finalize() = 0;
    virtual int32_t GetTrainStep() = 0;
};
IPCEnv* NewIPCEnvImpl(int32_t device_count, bool create_shm);

// ---------------------------------------------------------------------------------------------
struct OpParams {  // SS/engine/operator.h:4-17
    int device_id;
    hipStream_t stream;
    hipEvent_t event;
    void* memorypool;
    void* cache;
    void* graph;
    void* feature;
    void* env;
    int neighbor_count;
    bool is_presc;
    bool in_memory;
    int hop_num;
};

class Operator {
public:
    virtual ~Operator() = default;
    virtual void run(OpParams* params) = 0;
};
Operator* NewBatchGenerateOP(int op_id);
Operator* NewRandomSampleOP(int op_id);
Operator* NewCacheLookupOP(int op_id);
Operator* NewSSDIOSubmitOP(int op_id);
Operator* NewSSDIOCompleteOP(int op_id);

struct RunnerParams {  // SS/engine/server.h:5-14
    int device_id;
    std::vector<int> fanout;
    void* cache;
    void* graph;
    void* feature;
    void* env;
    int global_batch_id;
    bool in_memory;
};

class Server {
public:
    virtual ~Server() = default;
    virtual void Initialize(int global_shard_count, std::vector<int> fanout, int in_memory_mode) = 0;
    virtual void PreSc(int cache_agg_mode) = 0;
    virtual void Run() = 0;
    virtual void Finalize() = 0;
};

class Runner {
public:
    virtual ~Runner() = default;
    virtual void Initialize(RunnerParams* params) = 0;
    virtual void InitializeFeaturesBuffer(RunnerParams* params) = 0;
    virtual void RunPreSc(RunnerParams* params) = 0;
    virtual void RunOnce(RunnerParams* params) = 0;
    virtual void Finalize(RunnerParams* params) = 0;
    virtual void PrepareServing(RunnerParams*) {}   // new: allocations / graph capture of the serving phase, before "ready"
};
Runner* NewGPURunner();

// one process per GPU with a clique spread over processes: only device `dev` is owned by this process
extern "C" void legion_set_local_device(int32_t dev);
bool lg_is_local(int32_t dev);

void lg_pool_alloc_private(MemoryPool* mp, int32_t dev_id, int32_t total_num_nodes, int32_t batch_size,
                           const int32_t* fanout, int32_t hop_num, int32_t float_feature_len);
// how many pools of this shape the caller is about to keep in flight on the device (Pipeline: lanes x slots);
// feeds the direct-vs-table choice of the position state (LEGION_DEDUP=auto).  Thread-local; 0 = one pool.
// what PreSC saw of the LAST hop, the largest one: its edges (= the claims its de-duplication takes) and the batch's nodes
// before it (= what that de-duplication must recognise), maxima over the PreSC batches; 0, 0 = unknown.  Pools created
// afterwards by this thread pick the small class's bucket count from it (8, or 16 where a bucket would need two passes).
void lg_set_pool_claims_hint(int64_t last_hop_edges, int64_t nodes_before_last_hop);

// Where the pools a thread is about to create put their trainer-visible arrays (sampled_ids, features, labels, agg_src_off,
// agg_dst_off, the two counter blocks).  Default (null): one allocation each.  GPURunner sets an arena -- ONE exportable device
// allocation for every lane of its groups, so that a trainer end can open it with a single IPC handle and take a batch as
// views of its lane (no copy into a pipe slot) -- and a host-visible block for the lanes' counters.  Thread-local.
struct PoolArena {
    char* base = nullptr;
    int64_t bytes = 0, used = 0;
    int32_t* mirror_host = nullptr;    // [mirror_lanes][32] mapped pinned host memory ...
    int32_t* mirror_dev = nullptr;     // ... and its device address
    int32_t mirror_lanes = 0, mirror_used = 0;
};
void lg_set_pool_arena(PoolArena* arena);
PoolArena* lg_get_pool_arena();
int64_t lg_pool_arena_bytes(int64_t batch_size, int64_t num_ids, int64_t feature_rows, int64_t float_feature_len);

// alloc helpers, SS/engine/server_imp.cuh:2-51
extern "C" void* d_alloc_space(int64_t num_bytes);
extern "C" void d_free_space(void* d_ptr);
extern "C" void* host_alloc_space(int64_t num_bytes);
extern "C" void SetGPUDevice(int32_t shard_id);
extern "C" int32_t GetGPUDevice();
// hipIpcGetMemHandle with a few retries: on this pool's driver (dmabuf IPC) an export now and then fails with
// 'invalid argument' for a fresh, valid allocation and succeeds a moment later; a persistent failure is fatal as before
void lg_ipc_export(void* handle64, void* dev_ptr, const char* file, int line);
void* lg_alloc_exported(int64_t num_bytes, void* handle64, const char* file, int line);   // fresh buffer + its handle; tries other blocks

// ---------------------------------------------------------------------------------------------
// kernel launchers (kernels_*.hip)
namespace lg {

// the process-wide LegionTuning (include/legion_hip.h section 6, tuning.hip): launch paths read it here, nothing else
// parses LEGION_* tuning variables.  tuning_refresh() re-reads the environment (unless a host program installed its own
// values with legion_tuning_set); pools, pipelines and servers call it when they are created.
LegionTuning tuning();           // a snapshot by value (tuning.hip)
void tuning_refresh();

struct HopParams {                  // what every lane of a launch shares
    int32_t op_id;
    int32_t count;                  // fan-out of this hop
    int32_t partition_count;        // P: slot of the full CSR in the pointer tables
    int32_t* const* csr_dst_node_ids;   // device table [P+1] of column arrays
    int32_t* const* csr_dst_x;          // device table [P+1] of {id, feature-cache slot} pair arrays (column slots), or null
    const int32_t* col_full;            // the full CSR's column array and its pair copy BY VALUE: a pick from slot P (nearly all of
    const int32_t* colx_full;           // them) then needs no dependent load of a table entry before the column load itself
    const RowHdr* row_hdr;          // [N] per-vertex row headers of this GPU
    bool last_hop;                  // no next hop: scatter skips the header lookup
    bool is_presc;
    int32_t max_slots;              // capacity of slot_dst for this hop
    unsigned long long* edge_access_time;  // presample only (single lane), else null
    unsigned long long* topo_transactions; // presample only: 64-byte transactions the hop's topology reads amount to
    int32_t lds_bucket_bits;        // LG_LDS_BITS_SMALL / SMALL16 / MEDIUM / LARGE (the pool's)
    int32_t lds_k;                  // super tiles per partition tile in this hop (set by launch_random_sample)
    int32_t dedup_claims;           // 64-bucket class: claims per thread the (last) hop's de-duplication keeps in registers: LG_DEDUP_CLAIMS, _MID or _BIG by what PreSC saw
};
void launch_random_sample(hipStream_t s, const HopParams& p, const LanePtrs* d_lanes, int32_t n_lanes);

// hand-over of a lane's finished batch to a trainer-visible pipe slot (kernels_gather.hip)
struct DeliverParams {
    int32_t* sampled_ids; int32_t* labels; int32_t* agg_src_off; int32_t* agg_dst_off;   // the slot's buffers
    int32_t* node_counter; int32_t* edge_counter;
    int32_t* mirror;          // device address of the slot's host-visible counter mirror [32], or null
    int32_t num_ids;          // capacity of the id / edge arrays
    int32_t batch_cap;        // capacity of labels
};

struct GatherParams {
    const float* full_table;
    const float* replica;           // local copy of the clique's `replica_rows` hottest rows (hotness rank order), or null
    int32_t replica_rows;
    int32_t Kg;                     // GPUs per clique: rank t of a hit = (g % cap) * Kg + g / cap
    int32_t member;                 // this GPU's index inside its clique (dev_id % Kg)
    bool striped;                   // slots encode (owner, row) = (g / capacity, g % capacity); false: one table, row = g (no division)
    unsigned long long* stats;      // optional {rows read through a stripe pointer, rows from the local replica, rows from a PEER's stripe}
    const float* const* cache_tables;
    const float* local_table;       // this member's own stripe by value (cache_tables[member]): no dependent pointer load for local rows
    const int32_t* node_map;
    int32_t node_capacity;
    int32_t D;
    int32_t total_num_nodes;
    int32_t max_rows;               // rows any lane can have for this op: the kernel never gathers more (and the grid covers them, unless ...)
    int32_t grid_rows;              // ... > 0: the rows a lane TYPICALLY has: the grid is sized for these (the workgroups walk a lane's tiles, so a
                                    // lane with more rows -- up to max_rows -- is still gathered whole); 0: size the grid for max_rows
    int32_t hop;                    // >= 0: take the range from hop_scratch[HS_RANGE + 2*hop] (snapshot that later
                                    // hops do not overwrite, so the gather may run beside the next hop); < 0: node_counter[0..1]
    int32_t first_hop;              // with hop >= 0: also gather the ranges of hops first_hop .. hop-1 (they are adjacent in
                                    // sampled_ids); == hop for a plain single-op gather
    bool last_op;                   // the gather of the batch's last op (kernel instance of its own: hand-over, traces)
    bool skip_remote;               // peer_gather = bulk: rows of OTHER members' stripes are not fetched here (their owners push them)
    // hybrid CPU-cache / GPU-cache tier (feat_cache_lookup, SS/cache/cache_impl.cuh:202-235): slots [0, cpu_cap) are rows of
    // hybrid_cpu_cache (mapped pinned host memory), slots >= cpu_cap rows (g - cpu_cap) % gpu_cap of local_table; a miss reads
    // full_table when that is bound and is left unwritten otherwise
    bool hybrid;
    int32_t hybrid_cpu_cap, hybrid_gpu_cap;
    const float* hybrid_cpu_cache;
};

// Owner-bucketed bulk transfer of a striped gather (LegionTuning.peer_gather = bulk; SURVEY section 7 "hard parts", the
// alternative to 512-1024-byte direct loads over xGMI, SS/cache/cache_impl.cuh:268).  The REQUESTER lists, per owner, the rows
// of that owner's stripe its launch group needs and where they go: row index inside the stripe, byte offset of the destination
// row inside the requester's lane arena.  The OWNER then reads its own HBM and pushes whole rows to the requester with
// coalesced posted stores (bulk_push_kernel).  One list set per pipeline slot, in requester memory, exported to the owners.
struct BulkLists {
    int32_t Kg;
    int64_t cap;                     // entries per owner list
    int32_t* fidx;                   // [Kg][cap]
    int64_t* dst;                    // [Kg][cap]
    unsigned long long* cnt;         // [Kg] entries listed (never beyond cap: a group has at most cap rows)
};
void launch_bulk_bucket(hipStream_t s, const GatherParams& g, const LanePtrs* d_lanes, int32_t n_lanes, const BulkLists& lists,
                        const char* arena_base);
void launch_bulk_push(hipStream_t s, const float* stripe, int32_t D, const int32_t* fidx, const int64_t* dst,
                      const unsigned long long* cnt, int64_t cap, char* peer_arena);
void launch_gather(hipStream_t s, const GatherParams& g, const LanePtrs* d_lanes, int32_t n_lanes);
void launch_deliver(hipStream_t s, const LanePtrs* d_lane, const DeliverParams& d);
// the whole finished batch of a lane -- ids, feature rows, labels, both edge arrays, counters -- copied into a pipe slot
// stand-alone form for tests / probes: explicit arrays, one lane
void launch_gather_explicit(hipStream_t s, const GatherParams& g, const int32_t* sampled_ids,
                            int32_t* cache_index_out, const int32_t* range, float* dst, int32_t dst_rows);

struct SeedParams {
    int32_t batch_size;
    int32_t counter0;               // lane i takes iteration counter0 + i (or iter_state[0] + i)
    const int32_t* all_ids;
    const int32_t* all_labels;
    int32_t total_cap;
    int32_t hop_num;
    const int32_t* iter_state;      // device {next iteration, stride} or null
};
void launch_batch_generate(hipStream_t s, const SeedParams& p, const LanePtrs* d_lanes, int32_t n_lanes);

void launch_end_of_batch(hipStream_t s, const LanePtrs* d_lanes, int32_t n_lanes, int32_t* iter_state);
void launch_hotness_measure(hipStream_t s, const int32_t* sampled_ids, const int32_t* node_counter,
                            unsigned long long* access_map);
void init_row_headers(hipStream_t s, RowHdr* hdr, const int64_t* csr_index, int32_t n, int32_t slot);
void cache_row_headers(hipStream_t s, RowHdr* hdr, const int32_t* QT, int32_t Kg, int32_t Ki, int32_t capacity,
                       int32_t n, const int64_t* d_index, int32_t slot);
void launch_find(hipStream_t s, const int32_t* keys, int32_t n, const int32_t* map32,
                 const char* map8, int32_t* out32, char* out8);
void launch_draw_batch(hipStream_t s, const int32_t* idx, const int32_t* deg, int32_t* out, int32_t n);

// a roctx range for the enclosing scope (markers.hip): visible to rocprofv3 --marker-trace, near-free otherwise
struct Range {
    explicit Range(const char* fmt, ...) __attribute__((format(printf, 2, 3)));
    ~Range();
    Range(const Range&) = delete;
    Range& operator=(const Range&) = delete;
private:
    bool on_;
};

// the hotness all-reduce (collective.hip): RCCL over the members of a clique that live in this process
bool clique_is_physical(const std::vector<int32_t>& devs);
double allreduce_u64_clique(const std::vector<int32_t>& devs, const std::vector<unsigned long long*>& send,
                            const std::vector<unsigned long long*>& recv, int64_t count);

// set-up kernels (kernels_cache.hip)
void aggregate_access(hipStream_t s, unsigned long long* agg, const unsigned long long* add, int32_t n);
void sort_hotness_desc(hipStream_t s, unsigned long long* keys_inout, int32_t* order_out, int32_t n);
void inclusive_scan_u64(hipStream_t s, const unsigned long long* in, unsigned long long* out, int32_t n);
void edge_mem_in_order(hipStream_t s, const int32_t* order, unsigned long long* edge_mem, int32_t n,
                       const int64_t* csr_index);
void init_node_map(hipStream_t s, int32_t* node_map, const int32_t* QF, int32_t capacity, int32_t Kg,
                   int32_t n);
void init_node_map_hybrid(hipStream_t s, int32_t* node_map, const int32_t* QF, int32_t cpu_cache_capacity,
                          int32_t gpu_cache_capacity, int32_t n);
void init_edge_maps(hipStream_t s, char* index_map, int32_t* offset_map, const int32_t* QT,
                    int32_t capacity, int32_t Kg, int32_t Ki, int32_t n);
void fill_value_i32(hipStream_t s, int32_t* p, int32_t v, int64_t n);
void build_column_slots(hipStream_t s, const int32_t* col, const int32_t* node_map, int32_t* colx_pairs, int64_t num_edges);
void fill_value_i8(hipStream_t s, char* p, char v, int64_t n);
void feat_fill_up(hipStream_t s, int32_t capacity, int32_t D, float* cache, const float* table,
                  const int32_t* QF, int32_t Kg, int32_t Ki, int32_t n);
void topo_neighbor_count(hipStream_t s, const int32_t* QT, int32_t Kg, int32_t Ki, int32_t capacity,
                         int32_t n, const int64_t* csr_index, int64_t* counts);
void inclusive_scan_i64(hipStream_t s, const int64_t* in, int64_t* out, int32_t n);
void topo_fill_up(hipStream_t s, const int32_t* QT, int32_t Kg, int32_t Ki, int32_t capacity, int32_t n,
                  const int64_t* csr_index, const int32_t* csr_dst, const int64_t* d_index,
                  int32_t* d_dst);
}  // namespace lg
