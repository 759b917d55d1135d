// cache.hip -- PreSCCacheController + UnifiedCache for the MI355X build.
//
// Reference: SS/cache/cache.cu (controller :4-291, UnifiedCache :295-748), SS/cache/cache.cuh.
// Same roles, same call order (Initialize -> InitializeCacheController -> [PreSC epoch:
// CacheProfiling] -> CandidateSelection -> CostModel -> FillUp -> Find*/FeatCacheLookup).
// What changed:
//   * the three BGHT cuckoo maps per GPU (cache.cu:80-86) are direct-mapped tables indexed by
//     vertex id (int32[N], int8[N], int32[N]) -- the find() contract (value or -2) is identical,
//     a lookup is one 4-byte read instead of up to three 128-byte bucket probes;
//   * hotness is reduced on the clique leader through peer pointers over xGMI exactly like
//     aggregate_access, or skipped when the caller already all-reduced it with RCCL
//     (one-process-per-GPU deployment);
//   * the hotness sort is rocPRIM's stable LSD radix sort (ties keep ascending vertex id, as the
//     reference's Thrust->CUB path does);
//   * MaxIdNum is tracked on the device (no blocking read-back per PreSC batch, cache.cu:55-61).
#include "legion_core.h"

#include <algorithm>
#include <atomic>
#include <iostream>

#define MIN_INTERVAL 0.01   // SS/cache/cache_impl.cuh:30
#define CLS 64              // SS/cache/cache_impl.cuh:31

namespace lg {
__global__ void find_range_kernel(const int32_t* __restrict__ sampled_ids,
                                  const int32_t* __restrict__ range, const int32_t* __restrict__ map,
                                  int32_t* __restrict__ out)
{
    const int32_t off = range[0], n = range[1];
    for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int32_t k = sampled_ids[off + i];
        out[i] = (k >= 0 && map) ? map[k] : CACHEMISS_FLAG;
    }
}

// max_ids[0]: nodes of a batch (cache.cu:55-61); [1], [2]: edges of its LAST hop and nodes before it -- the claims and the
// known vertices of the largest de-duplication (lg_set_pool_claims_hint)
__global__ void track_max_kernel(const int32_t* __restrict__ nc, const int32_t* __restrict__ ec, int32_t* __restrict__ max_ids)
{
    atomicMax(max_ids, nc[INTRABATCH_CON * 2 + 1]);
    const int32_t H = nc[INTRABATCH_CON * 3 - 1];
    if (ec != nullptr && H >= 1 && H <= 6) {
        atomicMax(max_ids + 1, ec[INTRABATCH_CON * 3 + H] - ec[INTRABATCH_CON * 3 + H - 1]);
        atomicMax(max_ids + 2, nc[INTRABATCH_CON * 3 + H - 1]);
    }
}
}  // namespace lg

class PreSCCacheController : public CacheController {
public:
    PreSCCacheController(int32_t train_step, int32_t device_count)
        : device_count_(device_count), train_step_(train_step) {}
    ~PreSCCacheController() override {}

    void Initialize(int32_t dev_id, int32_t total_num_nodes) override
    {
        device_idx_ = dev_id;
        total_num_nodes_ = total_num_nodes;
        SetGPUDevice(dev_id);
        const int64_t bytes = (int64_t)total_num_nodes * sizeof(unsigned long long);
        node_access_time_ = (unsigned long long*)d_alloc_space(bytes);
        // one extra word behind the edge counters: the PreSC epoch's topology transaction count
        edge_access_time_ = (unsigned long long*)d_alloc_space(bytes + (int64_t)sizeof(unsigned long long));
        HIP_CALL(hipMemset(node_access_time_, 0, (size_t)bytes));
        HIP_CALL(hipMemset(edge_access_time_, 0, (size_t)bytes + sizeof(unsigned long long)));
        d_max_ids_ = (int32_t*)d_alloc_space(16);
        HIP_CALL(hipMemset(d_max_ids_, 0, 16));
        iter_ = 0;
    }

    void Finalize() override
    {
        SetGPUDevice(device_idx_);
        d_free_space(node_access_time_);
        d_free_space(edge_access_time_);
        d_free_space(d_max_ids_);
        d_free_space(node_map_);
        d_free_space(edge_index_map_);
        d_free_space(edge_offset_map_);
        node_access_time_ = edge_access_time_ = nullptr;
        d_max_ids_ = node_map_ = edge_offset_map_ = nullptr;
        edge_index_map_ = nullptr;
    }

    // SS/cache/cache.cu:40-68
    void CacheProfiling(int32_t* sampled_ids, int32_t*, int32_t*, int32_t*, int32_t*,
                        int32_t* node_counter, int32_t* edge_counter, bool is_presc, void* stream) override
    {
        if (is_presc) {
            hipStream_t s = static_cast<hipStream_t>(stream);
            lg::launch_hotness_measure(s, sampled_ids, node_counter, node_access_time_);
            lg::track_max_kernel<<<1, 1, 0, s>>>(node_counter, edge_counter, d_max_ids_);
            hipCheckError();
            if (iter_ == (train_step_ - 1)) iter_ = 0;
        }
        iter_++;
    }

    // SS/cache/cache.cu:71-88: sentinel-filled tables instead of three hash maps
    void InitializeMap(int node_capacity, int edge_capacity) override
    {
        SetGPUDevice(device_idx_);
        node_capacity_ = node_capacity;
        edge_capacity_ = edge_capacity;
        d_free_space(node_map_);
        d_free_space(edge_index_map_);
        d_free_space(edge_offset_map_);
        node_map_ = (int32_t*)d_alloc_space((int64_t)total_num_nodes_ * 4);
        edge_index_map_ = (char*)d_alloc_space((int64_t)total_num_nodes_);
        edge_offset_map_ = (int32_t*)d_alloc_space((int64_t)total_num_nodes_ * 4);
        lg::fill_value_i32(nullptr, node_map_, CACHEMISS_FLAG, total_num_nodes_);
        lg::fill_value_i8(nullptr, edge_index_map_, (char)CACHEMISS_FLAG, total_num_nodes_);
        lg::fill_value_i32(nullptr, edge_offset_map_, CACHEMISS_FLAG, total_num_nodes_);
        HIP_CALL(hipDeviceSynchronize());
    }

    // SS/cache/cache.cu:90-136 (InitPair / InitIndexPair / InitOffsetPair + insert)
    void Insert(int32_t* QT, int32_t* QF, int32_t cache_expand, int32_t Kg) override
    {
        SetGPUDevice(device_idx_);
        (void)cache_expand;   // == Kg in every mode (cache.cu:555-564)
        lg::init_node_map(nullptr, node_map_, QF, node_capacity_, Kg, total_num_nodes_);
        lg::init_edge_maps(nullptr, edge_index_map_, edge_offset_map_, QT, edge_capacity_, Kg,
                           device_idx_ / Kg, total_num_nodes_);
        HIP_CALL(hipDeviceSynchronize());
    }

    // SS/cache/cache.cu:138-153 (HybridInitPair + insert; features only, the two topology maps stay empty)
    void HybridInsert(int32_t* QF, int32_t cpu_cache_capacity, int32_t gpu_cache_capacity) override
    {
        SetGPUDevice(device_idx_);
        lg::init_node_map_hybrid(nullptr, node_map_, QF, cpu_cache_capacity, gpu_cache_capacity, total_num_nodes_);
        HIP_CALL(hipDeviceSynchronize());
    }

    void AccessCount(int32_t*, int32_t, void*) override {}
    unsigned long long int* GetNodeAccessedMap() override { return node_access_time_; }
    unsigned long long int* GetEdgeAccessedMap() override { return edge_access_time_; }
    unsigned long long int* GetTopoTransactions() override { return edge_access_time_ ? edge_access_time_ + total_num_nodes_ : nullptr; }

    // SS/cache/cache.cu:180-215 -- range taken from the device counters, no read-back
    void FindFeat(int32_t* sampled_ids, int32_t* cache_offset, int32_t* node_counter, int32_t op_id,
                  void* stream) override
    {
        lg::find_range_kernel<<<1024, 256, 0, static_cast<hipStream_t>(stream)>>>(
            sampled_ids, node_counter + (op_id % INTRABATCH_CON) * 2, node_map_, cache_offset);
        hipCheckError();
    }

    // SS/cache/cache.cu:217-225
    void FindTopo(int32_t* input_ids, char* partition_index, int32_t* partition_offset,
                  int32_t batch_size, int32_t, void* strm_hdl, int32_t) override
    {
        lg::launch_find(static_cast<hipStream_t>(strm_hdl), input_ids, batch_size, edge_offset_map_,
                        edge_index_map_, partition_offset, partition_index);
    }

    int32_t MaxIdNum() override
    {
        SetGPUDevice(device_idx_);
        int32_t v = 0;
        HIP_CALL(hipMemcpy(&v, d_max_ids_, 4, hipMemcpyDeviceToHost));
        return v;
    }
    // (new in this build) what PreSC saw of the last hop: {its edges, the batch's nodes before it}, maxima over the batches
    void LastHopMax(int32_t out[2])
    {
        SetGPUDevice(device_idx_);
        HIP_CALL(hipMemcpy(out, d_max_ids_ + 1, 8, hipMemcpyDeviceToHost));
    }

    const int32_t* NodeMap() const override { return node_map_; }
    const char* EdgeIndexMap() const override { return edge_index_map_; }
    const int32_t* EdgeOffsetMap() const override { return edge_offset_map_; }

private:
    int32_t device_idx_ = 0;
    int32_t device_count_ = 0;
    int32_t total_num_nodes_ = 0;
    unsigned long long* node_access_time_ = nullptr;
    unsigned long long* edge_access_time_ = nullptr;
    int32_t train_step_ = 0;
    int32_t iter_ = 0;
    int32_t* d_max_ids_ = nullptr;
    int32_t* node_map_ = nullptr;
    char* edge_index_map_ = nullptr;
    int32_t* edge_offset_map_ = nullptr;
    int32_t node_capacity_ = 0, edge_capacity_ = 0;
};

CacheController* NewPreSCCacheController(int32_t train_step, int32_t device_count)
{
    return new PreSCCacheController(train_step, device_count);
}

// =============================================================================================
void UnifiedCache::Initialize(int64_t cache_memory, int32_t float_feature_len, int32_t train_step,
                              int32_t device_count, int32_t cpu_cache_capacity, int32_t gpu_cache_capacity)
{
    device_count_ = device_count;
    cache_controller_.resize(device_count_);
    for (int32_t i = 0; i < device_count_; i++)
        cache_controller_[i] = NewPreSCCacheController(train_step, device_count_);
    float_feature_cache_.assign(device_count_, nullptr);
    d_float_feature_cache_ptr_.assign(device_count_, nullptr);
    cache_memory_ = cache_memory;
    float_feature_len_ = float_feature_len;
    cpu_cache_capacity_ = cpu_cache_capacity;
    gpu_cache_capacity_ = gpu_cache_capacity;
    is_presc_ = true;
}

void UnifiedCache::InitializeCacheController(int32_t dev_id, int32_t total_num_nodes)
{
    total_num_nodes_ = total_num_nodes;
    cache_controller_[dev_id]->Initialize(dev_id, total_num_nodes);
}

void UnifiedCache::Finalize(int32_t dev_id)
{
    SetGPUDevice(dev_id);
    cache_controller_[dev_id]->Finalize();
    if (dev_id < (int32_t)cpu_cache_host_.size() && cpu_cache_host_[dev_id] != nullptr) {     // the hybrid tier's CPU cache
        legion_host_free(cpu_cache_host_[dev_id]);
        cpu_cache_host_[dev_id] = cpu_cache_dev_[dev_id] = nullptr;
    }
}

void UnifiedCache::FindFeat(int32_t* sampled_ids, int32_t* cache_offset, int32_t* node_counter,
                            int32_t op_id, void* stream, int32_t dev_id)
{
    cache_controller_[dev_id]->FindFeat(sampled_ids, cache_offset, node_counter, op_id, stream);
}

void UnifiedCache::FindTopo(int32_t* input_ids, char* partition_index, int32_t* partition_offset,
                            int32_t batch_size, int32_t op_id, void* strm_hdl, int32_t dev_id)
{
    cache_controller_[dev_id]->FindTopo(input_ids, partition_index, partition_offset, batch_size, op_id,
                                        strm_hdl, dev_id);
}

void UnifiedCache::CacheProfiling(int32_t* sampled_ids, int32_t* agg_src_id, int32_t* agg_dst_id,
                                  int32_t* agg_src_off, int32_t* agg_dst_off, int32_t* node_counter,
                                  int32_t* edge_counter, void* stream, int32_t dev_id)
{
    cache_controller_[dev_id]->CacheProfiling(sampled_ids, agg_src_id, agg_dst_id, agg_src_off,
                                              agg_dst_off, node_counter, edge_counter, is_presc_, stream);
}

// SS/cache/cache.cu:360-443
void UnifiedCache::CandidateSelection(int cache_agg_mode, FeatureStorage* feature, GraphStorage*)
{
    int32_t Kg = 1 << cache_agg_mode;                 // :375-389
    if (Kg > device_count_) Kg = device_count_;
    if (Kg < 1) Kg = 1;
    const int32_t Kc = device_count_ / Kg;
    Kc_ = Kc;
    Kg_ = Kg;
    cache_agg_mode_ = cache_agg_mode;
    const int32_t N = feature ? feature->TotalNodeNum() : total_num_nodes_;
    total_num_nodes_ = N;
    for (void* p : QF_) d_free_space(p);
    for (void* p : QT_) d_free_space(p);
    for (void* p : AF_) d_free_space(p);
    for (void* p : AT_) d_free_space(p);
    QF_.clear(); QT_.clear(); AF_.clear(); AT_.clear();

    hotness_reduce_path_.assign(Kc, 0);
    hotness_reduce_ms_ = 0;
    const int32_t how = lg::tuning().hotness_reduce;   // -1 auto, 0 the leader loop over peer pointers, 1 RCCL
    for (int32_t i = 0; i < Kc; i++) {
        int32_t lead = -1;                             // the clique leader; in a clique spread over processes
        for (int32_t j = 0; j < Kg && lead < 0; j++)   // every process leads with its own member
            if (lg_is_local(i * Kg + j)) lead = i * Kg + j;
        if (lead < 0) {
            QF_.push_back(nullptr); AF_.push_back(nullptr); QT_.push_back(nullptr); AT_.push_back(nullptr);
            continue;
        }
        // The clique sum of the counters (cache.cu:408-411,428-431).  North star: an RCCL all-reduce over xGMI -- every member
        // ends up with the sum, in place, and the leader sorts its own copy.  Possible when every member lives in this
        // process on a physical GPU of its own (the reference's deployment: one server, a thread per GPU); logical GPUs that
        // share a device (tests on one GPU) cannot form a communicator and keep the reference's leader loop.
        // (the all-reduce writes the sums into scratch arrays, not over the members' counters -- a second
        // CandidateSelection over the same counters (another cache_agg_mode, a re-fill after more presampling) then sums the
        // counters again, not sums of sums -- and an RCCL failure in `auto` falls back to the leader loop instead of ending the
        // process.)
        bool reduced = world_reduced;
        unsigned long long* summed[2] = {nullptr, nullptr};       // the leader's copy of the clique sums (RCCL path), else null
        if (!world_reduced) {
            std::vector<int32_t> devs;
            for (int32_t j = 0; j < Kg; j++)
                if (lg_is_local(i * Kg + j)) devs.push_back(i * Kg + j);
            const bool all_local = (int32_t)devs.size() == Kg;
            const bool distinct = all_local && lg::clique_is_physical(devs);
            if (how == 1 && !distinct) {
                printf("legion_hip: LEGION_HOTNESS_REDUCE=rccl, but the %d members of clique %d do not sit on %d distinct GPUs of this process\n", Kg, i, Kg);
                exit(EXIT_FAILURE);
            }
            bool rccl_ok = false;
            if (how == 1 || (how == -1 && distinct && Kg > 1)) {
                double ms = 0;
                rccl_ok = true;
                for (int which = 0; which < 2 && rccl_ok; which++) {
                    std::vector<unsigned long long*> send, recv;
                    for (int32_t d : devs) {
                        SetGPUDevice(d);
                        send.push_back(which == 0 ? cache_controller_[d]->GetNodeAccessedMap() : cache_controller_[d]->GetEdgeAccessedMap());
                        recv.push_back((unsigned long long*)d_alloc_space((int64_t)N * sizeof(unsigned long long)));
                    }
                    const double t = lg::allreduce_u64_clique(devs, send, recv, N);
                    rccl_ok = t >= 0;
                    ms += rccl_ok ? t : 0;
                    for (size_t j = 0; j < devs.size(); j++) {
                        SetGPUDevice(devs[j]);
                        if (rccl_ok && devs[j] == lead) summed[which] = recv[j];
                        else d_free_space(recv[j]);
                    }
                }
                if (!rccl_ok) {
                    for (int which = 0; which < 2; which++) { if (summed[which]) { SetGPUDevice(lead); d_free_space(summed[which]); summed[which] = nullptr; } }
                    if (how == 1) { printf("legion_hip: LEGION_HOTNESS_REDUCE=rccl and RCCL failed\n"); exit(EXIT_FAILURE); }
                    std::cout << "Hotness reduce on clique " << i << ": RCCL refused; falling back to the leader loop over peer pointers\n";
                } else {
                    std::cout << "Hotness reduce on clique " << i << ": RCCL all-reduce (ncclUint64, ncclSum) over " << Kg << " GPU" << (Kg > 1 ? "s" : "")
                              << ", 2 x " << N << " x 8 bytes per GPU, " << ms << " ms\n";
                    hotness_reduce_path_[i] = 2;
                    hotness_reduce_ms_ += ms;
                    reduced = true;
                }
            }
            if (!rccl_ok && Kg > 1) {
                std::cout << "Hotness reduce on clique " << i << ": leader loop over peer pointers ("
                          << (all_local ? (distinct ? "RCCL unavailable" : "its logical GPUs share physical devices") : "members in other processes") << ")\n";
                hotness_reduce_path_[i] = 1;
            }
        }
        SetGPUDevice(lead);
        for (int which = 0; which < 2; which++) {     // 0: node hotness -> QF/AF, 1: edge hotness -> QT/AT
            int32_t* order = (int32_t*)d_alloc_space((int64_t)N * sizeof(int32_t));
            unsigned long long* agg = summed[which];
            if (agg == nullptr) {
                agg = (unsigned long long*)d_alloc_space((int64_t)N * sizeof(unsigned long long));
                HIP_CALL(hipMemset(agg, 0, (size_t)N * sizeof(unsigned long long)));
                // aggregate_access on the leader reading each member's counters (peer loads over xGMI).
                // When the counters were already all-reduced across processes (RCCL), every member
                // holds the clique sum: take the leader's copy once.
                for (int32_t j = 0; j < Kg; j++) {
                    if (reduced ? (i * Kg + j != lead) : !lg_is_local(i * Kg + j)) continue;
                    CacheController* cc = cache_controller_[i * Kg + j];
                    lg::aggregate_access(nullptr, agg, which == 0 ? cc->GetNodeAccessedMap() : cc->GetEdgeAccessedMap(), N);
                }
            }
            lg::sort_hotness_desc(nullptr, agg, order, N);
            if (which == 0) { QF_.push_back(order); AF_.push_back(agg); }
            else { QT_.push_back(order); AT_.push_back(agg); }
        }
        HIP_CALL(hipDeviceSynchronize());
    }
    is_presc_ = false;
}

// SS/cache/cache.cu:445-551: the split of a clique's cache memory between feature rows and adjacency.  The budget is walked in granules of
// 1 % (MIN_INTERVAL); at every granule count k the transactions a topology cache of k granules and a feature cache of the remaining
// steps - 1 - k granules would save are added up, and the k with the largest sum wins.  The arithmetic TYPES are the reference's (float
// tables filled from double expressions, integer capacities stored through float: oracle/legion_oracle.c lgo_cost_model restates the
// same) because they decide which k wins on ties; prefix[-1] reads as 0; the two log lines are part of the launcher's contract.
void UnifiedCache::CostModel(int, FeatureStorage* feature, GraphStorage* graph,
                             std::vector<uint64_t>& counters, int32_t train_step)
{
    const int32_t N = feature->TotalNodeNum();
    const int32_t D = feature->GetFloatFeatureLen();
    const int64_t* csr_index = graph->GetCSRNodeIndexCPU();
    node_capacity_.clear();
    edge_capacity_.clear();
    for (int32_t i = 0; i < Kc_; i++) {
        if (QF_[i] == nullptr) { node_capacity_.push_back(0); edge_capacity_.push_back(0); continue; }
        for (int32_t j = 0; j < Kg_; j++)
            if (lg_is_local(i * Kg_ + j)) { SetGPUDevice(i * Kg_ + j); break; }
        const int transaction_bytes = CLS;
        const int64_t granule = (int64_t)((double)(cache_memory_ * Kg_) * MIN_INTERVAL);
        const uint64_t topo_tx_total = counters[0] + counters[1];
        uint64_t feat_tx_total = 0;
        for (int j = 0; j < Kg_; j++)     // the reference indexes controller j, not i*Kg+j (:462)
            feat_tx_total += (uint64_t)((int64_t)(((int64_t)(j < (int32_t)peer_max_ids_.size() ? peer_max_ids_[j]
                                                                   : cache_controller_[j]->MaxIdNum()) * train_step) * D) * sizeof(float)) /
                                   (uint64_t)transaction_bytes;

        unsigned long long* d_prefix = (unsigned long long*)d_alloc_space((int64_t)N * 8);
        unsigned long long* d_edge_mem = (unsigned long long*)d_alloc_space((int64_t)N * 8);
        std::vector<uint64_t> node_hot_prefix(N), edge_hot_prefix(N), adjacency_bytes_prefix(N);
        lg::inclusive_scan_u64(nullptr, AF_[i], d_prefix, N);
        HIP_CALL(hipMemcpy(node_hot_prefix.data(), d_prefix, (size_t)N * 8, hipMemcpyDeviceToHost));
        lg::inclusive_scan_u64(nullptr, AT_[i], d_prefix, N);
        HIP_CALL(hipMemcpy(edge_hot_prefix.data(), d_prefix, (size_t)N * 8, hipMemcpyDeviceToHost));
        lg::edge_mem_in_order(nullptr, QT_[i], d_edge_mem, N, csr_index);
        lg::inclusive_scan_u64(nullptr, d_edge_mem, d_prefix, N);
        HIP_CALL(hipMemcpy(adjacency_bytes_prefix.data(), d_prefix, (size_t)N * 8, hipMemcpyDeviceToHost));
        d_free_space(d_prefix);
        d_free_space(d_edge_mem);

        const int64_t clique_bytes = cache_memory_ * Kg_;
        if (granule <= 0) {
            printf("cache_memory %lld is too small for the cost model\n", (long long)cache_memory_);
            exit(EXIT_FAILURE);
        }
        const int64_t steps = (clique_bytes - 1) / granule + 1;
        int64_t step = 0;
        int32_t topo_vertices = 0, feat_rows = 0;
        std::vector<float> topo_tx_at(steps + 1, 0), feat_tx_at(steps + 1, 0);
        std::vector<float> topo_rows_at(steps + 1, 0), feat_rows_at(steps + 1, 0), saved_tx_at(steps + 1, 0);
        auto at = [](const std::vector<uint64_t>& v, int64_t k) -> uint64_t { return k < 0 ? 0 : v[k]; };
        for (int64_t budget = 0; budget < clique_bytes; budget += granule) {
            if ((uint64_t)budget > (uint64_t)N * D * sizeof(float))
                feat_rows = N;
            else
                feat_rows = (int32_t)((uint64_t)(step + 1) * ((uint64_t)granule / (D * sizeof(float))));
            if ((uint64_t)budget > adjacency_bytes_prefix[N - 1])
                topo_vertices = N;
            else
                topo_vertices = (int32_t)(std::lower_bound(adjacency_bytes_prefix.begin(), adjacency_bytes_prefix.end(),
                                                           (uint64_t)budget) - adjacency_bytes_prefix.begin());
            if (topo_vertices < N) {
                topo_tx_at[step] = (float)((double)topo_tx_total * 1.0 / (double)edge_hot_prefix[N - 1] *
                                                       (double)at(edge_hot_prefix, (int64_t)topo_vertices - 1));
                topo_rows_at[step] = (float)(topo_vertices / Kg_);
            }
            if (feat_rows < N) {
                feat_tx_at[step] = (float)((double)feat_tx_total * 1.0 / (double)node_hot_prefix[N - 1] *
                                                       (double)at(node_hot_prefix, (int64_t)feat_rows - 1));
                feat_rows_at[step] = (float)(feat_rows / Kg_);
            }
            step++;
        }
        for (int64_t k = 1; k < steps; k++)
            saved_tx_at[k] = topo_tx_at[k] + feat_tx_at[steps - 1 - k];
        const int64_t best = std::max_element(saved_tx_at.begin(), saved_tx_at.end()) - saved_tx_at.begin();
        std::cout << "Alpha: " << (best * MIN_INTERVAL) << " Transactions: " << saved_tx_at[best]
                  << " on Clique: " << i << std::endl;
        node_capacity_.push_back((int32_t)(feat_rows_at[steps - 1 - best] + 1));
        edge_capacity_.push_back((int32_t)(topo_rows_at[best] + 1));
        std::cout << "Feat capacity: " << feat_rows_at[steps - 1 - best] << " Topo capacity: "
                  << topo_rows_at[best] << " on Clique: " << i << std::endl;
    }
}

void UnifiedCache::SetCapacity(int32_t node_capacity, int32_t edge_capacity)
{
    node_capacity_.assign(Kc_, node_capacity);
    edge_capacity_.assign(Kc_, edge_capacity);
}

int32_t UnifiedCache::NodeCapacity(int32_t dev_id) const
{
    if (node_capacity_.empty()) return 0;
    return node_capacity_[dev_id / Kg_];
}

int32_t UnifiedCache::EdgeCapacity(int32_t dev_id) const
{
    if (edge_capacity_.empty()) return 0;
    return edge_capacity_[dev_id / Kg_];
}

static uint64_t next_cache_uid()
{
    static std::atomic<uint64_t> next_uid{1};
    return next_uid.fetch_add(1);
}

// SS/cache/cache.cu:553-611, in two steps so that a clique spread over processes can exchange its
// stripes in between (in one process FillUp = FillUpLocal + FillUpLink).
void UnifiedCache::FillUp(int cache_agg_mode, FeatureStorage* feature, GraphStorage* graph)
{
    (void)cache_agg_mode;
    FillUpLocal(feature, graph);
    FillUpLink(feature, graph);
}

void UnifiedCache::FillUpLocal(FeatureStorage* feature, GraphStorage* graph)
{
    {   // a new fill: pairs built from the previous node_map (of this or any other cache) are stale from here on
        if (uid_ == 0) uid_ = next_cache_uid();
        fill_generation_++;
        for (int32_t d = 0; d < device_count_; d++)
            if (lg_is_local(d)) graph->DropColumnSlots(d);
    }
    hybrid_ = false;                  // (a clique fill after HybridInit: the gather decodes (owner, row) slots again)
    const int32_t N = feature->TotalNodeNum();
    float* cpu_float_feature = feature->GetAllFloatFeature();
    cpu_float_features_ = cpu_float_feature;
    // (stripes and replica are dense.  Rows padded to whole 128-byte lines were built and measured in round 4 -- a row that is not a
    // whole number of lines costs the same lines at either pitch, 400 bytes at any 16-byte offset cover exactly four -- and removed:
    // DESIGN_HISTORY.md, profiles/r04/gather_pitch.md)
    const int32_t pitch = float_feature_len_;
    for (int32_t i = 0; i < Kc_; i++)
        for (int32_t j = 0; j < Kg_; j++) {
            const int32_t dev_id = i * Kg_ + j;
            if (!lg_is_local(dev_id) || QF_[i] == nullptr) continue;
            SetGPUDevice(dev_id);
            cache_controller_[dev_id]->InitializeMap(node_capacity_[i], edge_capacity_[i]);
            cache_controller_[dev_id]->Insert(QT_[i], QF_[i], Kg_, Kg_);
            d_free_space(d_float_feature_cache_ptr_[dev_id]);
            d_float_feature_cache_ptr_[dev_id] = (float**)d_alloc_space(device_count_ * sizeof(float*));
            if (float_feature_len_ > 0) {                      // this member's stripe: rows QF[r*Kg + j]
                d_free_space(float_feature_cache_[dev_id]);
                float* new_cache = (float*)d_alloc_space((int64_t)node_capacity_[i] * pitch * sizeof(float));
                lg::feat_fill_up(nullptr, node_capacity_[i], float_feature_len_, new_cache, cpu_float_feature,
                                 QF_[i], Kg_, j, N);
                HIP_CALL(hipDeviceSynchronize());
                float_feature_cache_[dev_id] = new_cache;
                // hot-row replica: ranks 0 .. R-1 of the clique order, identical on every member (Kg = 1 addressing)
                if ((int32_t)replica_.size() < device_count_) { replica_.resize(device_count_, nullptr); replica_rows_.resize(device_count_, 0); }
                d_free_space(replica_[dev_id]);
                replica_[dev_id] = nullptr;
                replica_rows_[dev_id] = 0;
                if (replica_bytes_ > 0 && Kg_ > 1) {
                    int64_t rows = replica_bytes_ / ((int64_t)pitch * sizeof(float));
                    rows = std::min<int64_t>(rows, std::min<int64_t>((int64_t)node_capacity_[i] * Kg_, N));
                    if (rows > 0) {
                        replica_[dev_id] = (float*)d_alloc_space(rows * pitch * sizeof(float));
                        lg::feat_fill_up(nullptr, (int32_t)rows, float_feature_len_, replica_[dev_id], cpu_float_feature, QF_[i], 1, 0, N);
                        HIP_CALL(hipDeviceSynchronize());
                        replica_rows_[dev_id] = (int32_t)rows;
                    }
                }
            }
        }
    for (int32_t i = 0; i < Kc_; i++)
        if (QT_[i] != nullptr) graph->GraphCacheBuildLocal(QT_[i], i, Kg_, edge_capacity_[i]);
}

void UnifiedCache::FillUpLink(FeatureStorage* feature, GraphStorage* graph)
{
    (void)feature;
    for (int32_t i = 0; i < Kc_; i++) {
        if (QF_[i] == nullptr) continue;
        std::vector<float*> table(device_count_, nullptr);     // indexed by in-clique GPU j (cache.cu:594)
        for (int32_t j = 0; j < Kg_; j++) table[j] = float_feature_cache_[i * Kg_ + j];   // local stripe or peer pointer
        for (int32_t j = 0; j < Kg_; j++) {                   // :598-601
            const int32_t dev_id = i * Kg_ + j;
            if (!lg_is_local(dev_id)) continue;
            SetGPUDevice(dev_id);
            HIP_CALL(hipMemcpy(d_float_feature_cache_ptr_[dev_id], table.data(), device_count_ * sizeof(float*),
                               hipMemcpyHostToDevice));
        }
        graph->GraphCacheLink(QT_[i], i, Kg_, edge_capacity_[i]);   // :606-608
    }
    for (int32_t i = 0; i < device_count_; i++) {
        if (!lg_is_local(i)) continue;
        SetGPUDevice(i);
        HIP_CALL(hipDeviceSynchronize());
        // column slots: with the id -> slot map of this GPU final, pair the column array with it (legion_core.h, GraphStorage)
        if (QF_[i / Kg_] != nullptr && float_feature_len_ > 0) graph->BuildColumnSlots(i, cache_controller_[i]->NodeMap(), FillStamp());
    }
}

// SS/cache/cache.cu:614-670.  Differences, all forced: the reference sorts each GPU's counters in place (here a copy: the
// counters survive), allocates ONE CPU cache on device 0 for every GPU although each GPU has its own order (here one per GPU),
// and fills neither cache (:616 allocates and nothing writes; :656 FeatFillUp is commented out) -- here both are filled from
// the table the way FillUp fills its stripes.  Ranks at or beyond N are skipped.
void UnifiedCache::HybridInit(FeatureStorage* feature, GraphStorage* graph, bool miss_from_table)
{
    const int32_t N = feature->TotalNodeNum();
    total_num_nodes_ = N;
    float* table = feature->GetAllFloatFeature();
    const int32_t cpu_cap = cpu_cache_capacity_ < 0 ? 0 : cpu_cache_capacity_;
    const int32_t gpu_cap = gpu_cache_capacity_ < 0 ? 0 : gpu_cache_capacity_;
    {   // a new fill (as in FillUpLocal): column slots built from an earlier node_map are stale
        if (uid_ == 0) uid_ = next_cache_uid();
        fill_generation_++;
        for (int32_t d = 0; d < device_count_; d++)
            if (lg_is_local(d)) graph->DropColumnSlots(d);
    }
    for (void* p : QF_) d_free_space(p);
    for (void* p : QT_) d_free_space(p);
    for (void* p : AF_) d_free_space(p);
    for (void* p : AT_) d_free_space(p);
    QF_.assign(device_count_, nullptr); AF_.assign(device_count_, nullptr);
    QT_.assign(device_count_, nullptr); AT_.assign(device_count_, nullptr);
    Kc_ = device_count_;              // every GPU on its own: "clique" i is GPU i
    Kg_ = 1;
    cache_agg_mode_ = 0;
    node_capacity_.assign(device_count_, gpu_cap + cpu_cap);
    edge_capacity_.assign(device_count_, 0);
    cpu_cache_dev_.resize(device_count_, nullptr);
    cpu_cache_host_.resize(device_count_, nullptr);
    for (int32_t i = 0; i < device_count_; i++) {
        if (!lg_is_local(i)) continue;
        SetGPUDevice(i);
        // :626-636 this GPU's counters alone, iota, sort_by_key(greater): ties keep ascending id
        unsigned long long* keys = (unsigned long long*)d_alloc_space((int64_t)N * sizeof(unsigned long long));
        HIP_CALL(hipMemcpy(keys, cache_controller_[i]->GetNodeAccessedMap(), (size_t)N * sizeof(unsigned long long), hipMemcpyDeviceToDevice));
        int32_t* order = (int32_t*)d_alloc_space((int64_t)N * sizeof(int32_t));
        lg::sort_hotness_desc(nullptr, keys, order, N);
        QF_[i] = order;
        AF_[i] = keys;
        cache_controller_[i]->InitializeMap(gpu_cap + cpu_cap, 100);     // :642 "edge cache disabled now"
        cache_controller_[i]->HybridInsert(order, cpu_cap, gpu_cap);     // :643
        graph->GraphCacheBuildLocal(order, i, 1, 0);                     // no cached topology: every row header back on the full CSR
        d_free_space(d_float_feature_cache_ptr_[i]);
        d_float_feature_cache_ptr_[i] = (float**)d_alloc_space(device_count_ * sizeof(float*));     // :647-652
        if (float_feature_len_ > 0) {
            d_free_space(float_feature_cache_[i]);
            float_feature_cache_[i] = (float*)d_alloc_space((int64_t)gpu_cap * float_feature_len_ * sizeof(float));   // :657
            if (cpu_cache_host_[i] != nullptr) legion_host_free(cpu_cache_host_[i]);
            void* host = nullptr;
            cpu_cache_dev_[i] = (float*)legion_host_alloc((int64_t)cpu_cap * float_feature_len_ * sizeof(float), &host);   // :616
            cpu_cache_host_[i] = (float*)host;
            lg::feat_fill_up(nullptr, std::min(gpu_cap, N), float_feature_len_, float_feature_cache_[i], table, order, 1, 0, N);
            if (N > gpu_cap)
                lg::feat_fill_up(nullptr, std::min(cpu_cap, N - gpu_cap), float_feature_len_, cpu_cache_dev_[i], table, order + gpu_cap, 1, 0,
                                 N - gpu_cap);
            HIP_CALL(hipDeviceSynchronize());
            std::vector<float*> tab(device_count_, nullptr);
            tab[0] = float_feature_cache_[i];
            HIP_CALL(hipMemcpy(d_float_feature_cache_ptr_[i], tab.data(), device_count_ * sizeof(float*), hipMemcpyHostToDevice));
        }
        if ((int32_t)replica_.size() > i && replica_[i] != nullptr) { d_free_space(replica_[i]); replica_[i] = nullptr; replica_rows_[i] = 0; }
    }
    hybrid_ = true;
    hybrid_miss_from_table_ = miss_from_table;
    hybrid_table_ = table;
    cpu_float_features_ = table;      // what the operators check to know the cache is bound (the reference points it at the CPU cache, :616)
    for (int32_t i = 0; i < device_count_; i++) {
        if (!lg_is_local(i)) continue;
        SetGPUDevice(i);
        if (float_feature_len_ > 0) graph->BuildColumnSlots(i, cache_controller_[i]->NodeMap(), FillStamp());
    }
    is_presc_ = false;
    std::cout << "Finish initializing cache\n";        // :669
}

int32_t UnifiedCache::MaxIdNum(int32_t dev_id) { return cache_controller_[dev_id]->MaxIdNum(); }
void UnifiedCache::LastHopMax(int32_t dev_id, int32_t out[2])
{
    out[0] = out[1] = 0;
    if (dev_id >= 0 && dev_id < (int32_t)cache_controller_.size() && cache_controller_[dev_id] != nullptr)
        static_cast<PreSCCacheController*>(cache_controller_[dev_id])->LastHopMax(out);
}

void UnifiedCache::SetGatherStats(bool on)
{
    gather_stats_on_ = on;
    const unsigned long long v = on ? 1ull : 0ull;
    for (size_t dvc = 0; dvc < gather_stats_.size(); dvc++) {
        if (gather_stats_[dvc] == nullptr) continue;
        SetGPUDevice((int32_t)dvc);
        HIP_CALL(hipDeviceSynchronize());                 // (launches in flight keep the value they started with)
        HIP_CALL(hipMemcpy(gather_stats_[dvc] + 3, &v, sizeof(v), hipMemcpyHostToDevice));
    }
}

unsigned long long* UnifiedCache::GatherStats(int32_t dev_id)
{
    if ((int32_t)gather_stats_.size() < device_count_) gather_stats_.resize(device_count_, nullptr);
    if (gather_stats_[dev_id] == nullptr) {
        SetGPUDevice(dev_id);
        gather_stats_[dev_id] = (unsigned long long*)d_alloc_space(4 * sizeof(unsigned long long));
        const unsigned long long init[4] = {0, 0, 0, gather_stats_on_ ? 1ull : 0ull};     // [3]: the device-side switch
        HIP_CALL(hipMemcpy(gather_stats_[dev_id], init, sizeof(init), hipMemcpyHostToDevice));
    }
    return gather_stats_[dev_id];
}

unsigned long long int* UnifiedCache::GetEdgeAccessedMap(int32_t dev_id)
{
    return cache_controller_[dev_id]->GetEdgeAccessedMap();
}

// SS/cache/cache.cu:726-748 -- lookup (FindFeat) fused into the gather
void UnifiedCache::FeatCacheLookup(const LanePtrs* d_lanes, int32_t n_lanes, int32_t op_id, int32_t dev_id,
                                   hipStream_t strm_hdl, int32_t max_rows, bool use_snapshot, int32_t first_op_id, bool last_op,
                                   bool skip_remote, int32_t grid_rows)
{
    lg::GatherParams g = GatherParamsOf(dev_id, op_id, max_rows, use_snapshot, first_op_id, last_op);
    g.skip_remote = skip_remote;
    g.grid_rows = grid_rows;
    lg::launch_gather(strm_hdl, g, d_lanes, n_lanes);
}

void UnifiedCache::BulkBucket(const LanePtrs* d_lanes, int32_t n_lanes, int32_t op_id, int32_t dev_id, hipStream_t s, int32_t max_rows,
                              const lg::BulkLists& lists, const char* arena_base)
{
    const lg::GatherParams g = GatherParamsOf(dev_id, op_id, max_rows, true, 1, true);
    lg::launch_bulk_bucket(s, g, d_lanes, n_lanes, lists, arena_base);
}

void UnifiedCache::BulkPush(int32_t owner_dev, hipStream_t s, const int32_t* fidx, const int64_t* dst, const unsigned long long* cnt,
                            int64_t cap, char* peer_arena)
{
    lg::launch_bulk_push(s, float_feature_cache_[owner_dev], float_feature_len_, fidx, dst, cnt, cap, peer_arena);
}

lg::GatherParams UnifiedCache::GatherParamsOf(int32_t dev_id, int32_t op_id, int32_t max_rows, bool use_snapshot, int32_t first_op_id,
                                              bool last_op)
{
    const bool filled = !node_capacity_.empty() && d_float_feature_cache_ptr_[dev_id] != nullptr &&
                        cache_controller_[dev_id]->NodeMap() != nullptr;
    (void)op_id;
    lg::GatherParams g;
    g.replica = (filled && dev_id < (int32_t)replica_.size()) ? replica_[dev_id] : nullptr;
    g.replica_rows = g.replica ? replica_rows_[dev_id] : 0;
    g.Kg = Kg_;
    g.member = dev_id % (Kg_ > 0 ? Kg_ : 1);
    g.striped = Kg_ > 1;
    g.stats = dev_id < (int32_t)gather_stats_.size() ? gather_stats_[dev_id] : nullptr;      // (armed or not: the kernel reads stats[3])
    g.full_table = cpu_float_features_;
    g.cache_tables = filled ? d_float_feature_cache_ptr_[dev_id] : nullptr;
    g.local_table = filled ? float_feature_cache_[dev_id] : nullptr;
    g.node_map = filled ? cache_controller_[dev_id]->NodeMap() : nullptr;
    g.node_capacity = filled ? NodeCapacity(dev_id) : 1;
    g.D = float_feature_len_;
    g.total_num_nodes = total_num_nodes_;
    g.max_rows = max_rows;
    g.grid_rows = 0;
    g.hop = use_snapshot ? op_id / INTRABATCH_CON : -1;
    g.first_hop = (use_snapshot && first_op_id >= 0 && first_op_id < op_id) ? first_op_id / INTRABATCH_CON : g.hop;
    g.last_op = last_op;
    g.skip_remote = false;
    g.hybrid = hybrid_ && filled;
    g.hybrid_cpu_cap = g.hybrid_gpu_cap = 0;
    g.hybrid_cpu_cache = nullptr;
    if (g.hybrid) {
        g.hybrid_cpu_cap = cpu_cache_capacity_ < 0 ? 0 : cpu_cache_capacity_;
        g.hybrid_gpu_cap = gpu_cache_capacity_ < 0 ? 0 : gpu_cache_capacity_;
        g.hybrid_cpu_cache = cpu_cache_dev_[dev_id];
        g.full_table = hybrid_miss_from_table_ ? hybrid_table_ : nullptr;
        g.striped = false;
        g.replica = nullptr;
        g.replica_rows = 0;
    }
    return g;
}

// ---- C API ----------------------------------------------------------------------------------
struct LegionCacheBox {   // the UnifiedCache plus what the C API needs to remember
    UnifiedCache cache;
    int32_t device_count = 0;
    int32_t total_num_nodes = 0;
};

static UnifiedCache* as_cache(LegionUnifiedCache* c) { return c ? &reinterpret_cast<LegionCacheBox*>(c)->cache : nullptr; }
static const UnifiedCache* as_cache(const LegionUnifiedCache* c) { return c ? &reinterpret_cast<const LegionCacheBox*>(c)->cache : nullptr; }

extern "C" void legion_cache_set_replica_memory(LegionUnifiedCache* c, int64_t bytes)
{
    if (UnifiedCache* u = as_cache(c)) u->SetReplicaMemory(bytes);
}
extern "C" int32_t legion_cache_replica_rows(const LegionUnifiedCache* c, int32_t dev_id)
{
    const UnifiedCache* u = as_cache(c);
    return u ? u->ReplicaRows(dev_id) : 0;
}
// enables the gather's row-source statistics for dev_id and returns {rows read from a peer's stripe, rows read from the
// local replica} counted so far (device counters, read back here)
extern "C" void legion_cache_gather_stats(LegionUnifiedCache* c, int32_t dev_id, uint64_t* out2)
{
    UnifiedCache* u = as_cache(c);
    if (!u) return;
    unsigned long long* d = u->GatherStats(dev_id);
    unsigned long long h[2] = {0, 0};
    HIP_CALL(hipDeviceSynchronize());
    HIP_CALL(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
    if (out2) { out2[0] = h[0]; out2[1] = h[1]; }
}

// counting costs the gather an atomic per hit row: a measurement switches it off again before anything is timed
extern "C" void legion_cache_gather_stats_enable(LegionUnifiedCache* c, int32_t on)
{
    if (UnifiedCache* u = as_cache(c)) u->SetGatherStats(on != 0);
}

extern "C" void legion_cache_gather_stats3(LegionUnifiedCache* c, int32_t dev_id, uint64_t* out3)
{
    UnifiedCache* u = as_cache(c);
    if (!u) return;
    unsigned long long* d = u->GatherStats(dev_id);
    unsigned long long h[3] = {0, 0, 0};
    HIP_CALL(hipDeviceSynchronize());
    HIP_CALL(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
    if (out3) { out3[0] = h[0]; out3[1] = h[1]; out3[2] = h[2]; }
}

// rows read from other members' stripes so far, as 64-byte transactions (the unit of CostModel's counters)
extern "C" uint64_t legion_cache_peer_transactions(LegionUnifiedCache* c, int32_t dev_id)
{
    UnifiedCache* u = as_cache(c);
    if (!u) return 0;
    uint64_t h[3] = {0, 0, 0};
    legion_cache_gather_stats3(c, dev_id, h);
    return h[2] * (uint64_t)u->FloatFeatureLen() * sizeof(float) / 64;
}

extern "C" LegionUnifiedCache* legion_cache_create(int64_t cache_memory, int32_t float_feature_len,
                                                   int32_t train_step, int32_t device_count,
                                                   int32_t total_num_nodes)
{
    LegionCacheBox* b = new LegionCacheBox();
    b->device_count = device_count;
    b->total_num_nodes = total_num_nodes;
    b->cache.Initialize(cache_memory, float_feature_len, train_step, device_count, 0, 0);
    b->cache.total_num_nodes_ = total_num_nodes;
    return reinterpret_cast<LegionUnifiedCache*>(b);
}

extern "C" void legion_cache_init_controller(LegionUnifiedCache* c, int32_t dev_id)
{
    if (!c) { printf("invalid cache ptr\n"); return; }
    LegionCacheBox* b = reinterpret_cast<LegionCacheBox*>(c);
    b->cache.InitializeCacheController(dev_id, b->total_num_nodes);
}

extern "C" void legion_cache_candidate_selection(LegionUnifiedCache* c, int32_t cache_agg_mode,
                                                 LegionGraphStorage* graph, int32_t world_reduced)
{
    UnifiedCache* u = as_cache(c);
    if (!u) { printf("invalid cache ptr\n"); return; }
    u->world_reduced = world_reduced != 0;
    u->CandidateSelection(cache_agg_mode, nullptr, reinterpret_cast<GraphStorage*>(graph));
}

// One process per GPU: all-reduce (RCCL, the process-wide communicator of legion_collective_init_rank) of GPU dev_id's two
// access-counter arrays in place; returns the world size the collective ran over (0: no communicator).  Afterwards call
// legion_cache_candidate_selection(..., world_reduced = 1).
extern "C" int32_t legion_collective_allreduce_u64(void* devptr, int64_t count, double* ms_out);
extern "C" int32_t legion_cache_allreduce_hotness(LegionUnifiedCache* c, int32_t dev_id, double* ms_out)
{
    UnifiedCache* u = as_cache(c);
    if (!u || !u->Controller(dev_id)) { printf("invalid cache ptr\n"); return 0; }
    double ms = 0, total = 0;
    int32_t world = legion_collective_allreduce_u64(u->Controller(dev_id)->GetNodeAccessedMap(), u->total_num_nodes_, &ms);
    total += ms;
    if (world > 0) world = legion_collective_allreduce_u64(u->Controller(dev_id)->GetEdgeAccessedMap(), u->total_num_nodes_, &ms);
    total += ms;
    if (ms_out) *ms_out = total;
    return world;
}
// how the last candidate selection summed the counters of dev_id's clique: 0 nothing to sum / as given, 1 leader loop over
// peer pointers, 2 RCCL all-reduce inside the library
extern "C" int32_t legion_cache_hotness_reduce_path(const LegionUnifiedCache* c, int32_t dev_id)
{
    const UnifiedCache* u = as_cache(c);
    if (!u || u->hotness_reduce_path_.empty()) return 0;
    const int32_t clique = dev_id / (u->Kg_ > 0 ? u->Kg_ : 1);
    return clique < (int32_t)u->hotness_reduce_path_.size() ? u->hotness_reduce_path_[clique] : 0;
}

extern "C" void legion_cache_cost_model(LegionUnifiedCache* c, LegionFeatureStorage* feature,
                                        LegionGraphStorage* graph, const uint64_t* counters, int32_t train_step)
{
    UnifiedCache* u = as_cache(c);
    if (!u || !feature || !graph) { printf("invalid cache/feature/graph ptr\n"); return; }
    std::vector<uint64_t> cnt(2, 0);
    if (counters) { cnt[0] = counters[0]; cnt[1] = counters[1]; }
    u->CostModel(u->cache_agg_mode_, reinterpret_cast<FeatureStorage*>(feature),
                 reinterpret_cast<GraphStorage*>(graph), cnt, train_step);
}

extern "C" void legion_cache_set_capacity(LegionUnifiedCache* c, int32_t node_capacity, int32_t edge_capacity)
{
    UnifiedCache* u = as_cache(c);
    if (u) u->SetCapacity(node_capacity, edge_capacity);
}

extern "C" void legion_cache_fill_up(LegionUnifiedCache* c, LegionFeatureStorage* feature, LegionGraphStorage* graph)
{
    UnifiedCache* u = as_cache(c);
    if (!u || !feature || !graph) { printf("invalid cache/feature/graph ptr\n"); return; }
    u->FillUp(u->cache_agg_mode_, reinterpret_cast<FeatureStorage*>(feature), reinterpret_cast<GraphStorage*>(graph));
}

// The hybrid CPU-cache / GPU-cache tier instead of candidate_selection + cost_model + fill_up (UnifiedCache::HybridInit)
extern "C" void legion_cache_hybrid_init(LegionUnifiedCache* c, LegionFeatureStorage* feature, LegionGraphStorage* graph,
                                         int32_t cpu_cache_capacity, int32_t gpu_cache_capacity, int32_t miss_from_table)
{
    UnifiedCache* u = as_cache(c);
    if (!u || !feature || !graph) { printf("invalid cache/feature/graph ptr\n"); return; }
    u->SetHybridCapacity(cpu_cache_capacity, gpu_cache_capacity);
    u->HybridInit(reinterpret_cast<FeatureStorage*>(feature), reinterpret_cast<GraphStorage*>(graph), miss_from_table != 0);
}

// device address of GPU dev_id's CPU cache (mapped pinned host memory, cpu_cache_capacity x D floats), or null
extern "C" const float* legion_cache_hybrid_cpu_cache(const LegionUnifiedCache* c, int32_t dev_id)
{
    const UnifiedCache* u = as_cache(c);
    return u ? u->HybridCPUCache(dev_id) : nullptr;
}

extern "C" const float* legion_cache_feature_cache(const LegionUnifiedCache* c, int32_t dev_id)
{
    const UnifiedCache* u = as_cache(c);
    return u ? u->FeatureCachePtr(dev_id) : nullptr;
}

extern "C" void legion_cache_destroy(LegionUnifiedCache* c)
{
    if (!c) return;
    LegionCacheBox* b = reinterpret_cast<LegionCacheBox*>(c);
    for (int32_t i = 0; i < b->device_count; i++)
        if (b->cache.Controller(i) && lg_is_local(i)) b->cache.Finalize(i);
    delete b;
}

extern "C" int32_t legion_cache_node_capacity(const LegionUnifiedCache* c, int32_t dev_id)
{
    const UnifiedCache* u = as_cache(c);
    return u ? u->NodeCapacity(dev_id) : 0;
}

extern "C" int32_t legion_cache_edge_capacity(const LegionUnifiedCache* c, int32_t dev_id)
{
    const UnifiedCache* u = as_cache(c);
    return u ? u->EdgeCapacity(dev_id) : 0;
}

// 64-byte transactions the topology reads of GPU dev_id's PreSC epoch amount to (the quantity the
// paper measured with Intel PCM on the PCIe root ports and v2 hard-wires to 0, SS/engine/server.cu:105-110):
// per sampled row one transaction for the row-pointer pair plus min(fan-out, ceil(4*deg/64)) for the picks.
extern "C" uint64_t legion_cache_topo_transactions(LegionUnifiedCache* c, int32_t dev_id)
{
    UnifiedCache* u = as_cache(c);
    if (!u) return 0;
    CacheController* cc = u->Controller(dev_id);
    if (!cc || !cc->GetTopoTransactions()) return 0;
    SetGPUDevice(dev_id);
    unsigned long long v = 0;
    HIP_CALL(hipDeviceSynchronize());
    HIP_CALL(hipMemcpy(&v, cc->GetTopoTransactions(), sizeof(v), hipMemcpyDeviceToHost));
    return (uint64_t)v;
}

extern "C" int32_t legion_cache_max_id_num(const LegionUnifiedCache* c, int32_t dev_id)
{
    UnifiedCache* u = as_cache(const_cast<LegionUnifiedCache*>(c));
    return u ? u->MaxIdNum(dev_id) : 0;
}

extern "C" void* legion_cache_array(LegionUnifiedCache* c, int32_t dev_id, int32_t which)
{
    UnifiedCache* u = as_cache(c);
    if (!u) return nullptr;
    const int32_t clique = dev_id / (u->Kg_ > 0 ? u->Kg_ : 1);
    CacheController* cc = u->Controller(dev_id);
    switch (which) {
        case 0: return clique < (int32_t)u->QF_.size() ? u->QF_[clique] : nullptr;
        case 1: return clique < (int32_t)u->QT_.size() ? u->QT_[clique] : nullptr;
        case 2: return clique < (int32_t)u->AF_.size() ? u->AF_[clique] : nullptr;
        case 3: return clique < (int32_t)u->AT_.size() ? u->AT_[clique] : nullptr;
        case 4: return cc->GetNodeAccessedMap();
        case 5: return cc->GetEdgeAccessedMap();
        case 6: return const_cast<int32_t*>(cc->NodeMap());
        case 7: return const_cast<char*>(cc->EdgeIndexMap());
        case 8: return const_cast<int32_t*>(cc->EdgeOffsetMap());
        default: return nullptr;
    }
}

// UnifiedCache::FindTopo / FindFeat as plain calls (SS/cache/cache.cu:335-357): device arrays in and out
extern "C" void legion_cache_find_topo(LegionUnifiedCache* c, int32_t dev_id, legion_stream_t stream,
                                       const int32_t* input_ids, int32_t batch_size, char* partition_index,
                                       int32_t* partition_offset)
{
    UnifiedCache* u = as_cache(c);
    if (!u) { printf("invalid cache ptr\n"); return; }
    u->FindTopo(const_cast<int32_t*>(input_ids), partition_index, partition_offset, batch_size, 0, stream, dev_id);
}

extern "C" void legion_cache_find_feat(LegionUnifiedCache* c, int32_t dev_id, legion_stream_t stream,
                                       const int32_t* sampled_ids, int32_t* cache_offset,
                                       const int32_t* node_counter, int32_t op_id)
{
    UnifiedCache* u = as_cache(c);
    if (!u) { printf("invalid cache ptr\n"); return; }
    u->FindFeat(const_cast<int32_t*>(sampled_ids), cache_offset, const_cast<int32_t*>(node_counter), op_id, stream, dev_id);
}

// ---- a clique spread over processes (one process per GPU): exchange of the stripes ---------------
// Order on every rank: [PreSC] -> all-reduce hotness -> legion_cache_set_peer_max_ids ->
// candidate_selection(world_reduced) -> cost_model -> legion_cache_fill_up_local -> legion_cache_export ->
// all-gather the handles -> legion_cache_import_peer for every other rank -> legion_cache_fill_up_link.
struct LegionStripeHandles {          // what one member publishes: 3 IPC handles
    hipIpcMemHandle_t feat_cache, topo_index, topo_col;
};
static_assert(sizeof(LegionStripeHandles) == 192, "three 64-byte IPC handles");

extern "C" void legion_cache_set_peer_max_ids(LegionUnifiedCache* c, const int32_t* max_ids, int32_t n)
{
    UnifiedCache* u = as_cache(c);
    if (u) u->SetPeerMaxIds(max_ids, n);
}

extern "C" void legion_cache_fill_up_local(LegionUnifiedCache* c, LegionFeatureStorage* feature, LegionGraphStorage* graph)
{
    UnifiedCache* u = as_cache(c);
    if (!u || !feature || !graph) { printf("invalid cache/feature/graph ptr\n"); return; }
    u->FillUpLocal(reinterpret_cast<FeatureStorage*>(feature), reinterpret_cast<GraphStorage*>(graph));
}

extern "C" void legion_cache_fill_up_link(LegionUnifiedCache* c, LegionFeatureStorage* feature, LegionGraphStorage* graph)
{
    UnifiedCache* u = as_cache(c);
    if (!u || !feature || !graph) { printf("invalid cache/feature/graph ptr\n"); return; }
    u->FillUpLink(reinterpret_cast<FeatureStorage*>(feature), reinterpret_cast<GraphStorage*>(graph));
}

extern "C" void legion_cache_export(LegionUnifiedCache* c, LegionGraphStorage* graph, int32_t dev_id, void* handles192)
{
    UnifiedCache* u = as_cache(c);
    GraphStorage* g = reinterpret_cast<GraphStorage*>(graph);
    if (!u || !g || !handles192) { printf("invalid cache/graph ptr\n"); return; }
    SetGPUDevice(dev_id);
    LegionStripeHandles* h = reinterpret_cast<LegionStripeHandles*>(handles192);
    lg_ipc_export(&h->feat_cache, (void*)u->FeatureCachePtr(dev_id), __FILE__, __LINE__);
    lg_ipc_export(&h->topo_index, (void*)g->CachedCSRIndex(dev_id), __FILE__, __LINE__);
    lg_ipc_export(&h->topo_col, (void*)g->CachedCSRDst(dev_id), __FILE__, __LINE__);
}

extern "C" void legion_cache_import_peer(LegionUnifiedCache* c, LegionGraphStorage* graph, int32_t local_dev,
                                         int32_t peer_dev, const void* handles192)
{
    UnifiedCache* u = as_cache(c);
    GraphStorage* g = reinterpret_cast<GraphStorage*>(graph);
    if (!u || !g || !handles192) { printf("invalid cache/graph ptr\n"); return; }
    SetGPUDevice(local_dev);
    const LegionStripeHandles* h = reinterpret_cast<const LegionStripeHandles*>(handles192);
    void *feat = nullptr, *ti = nullptr, *tc = nullptr;
    HIP_CALL(hipIpcOpenMemHandle(&feat, h->feat_cache, hipIpcMemLazyEnablePeerAccess));
    HIP_CALL(hipIpcOpenMemHandle(&ti, h->topo_index, hipIpcMemLazyEnablePeerAccess));
    HIP_CALL(hipIpcOpenMemHandle(&tc, h->topo_col, hipIpcMemLazyEnablePeerAccess));
    u->SetPeerFeatureCache(peer_dev, (float*)feat);
    g->SetPeerCSR(peer_dev, (int64_t*)ti, (int32_t*)tc);
}
