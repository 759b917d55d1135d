// kernels_cache.hip -- one-time set-up kernels that feed the hot path: hotness reduction,
// stable descending hotness sort, prefix sums, id->slot tables, cache fills.
//
// Reference (SS = sampling_server/src):
//   aggregate_access / init_cache_order      SS/cache/cache_impl.cuh:72-83
//   thrust::sort_by_key(greater<u64>)        SS/cache/cache.cu:415,435   -> rocPRIM radix sort (stable)
//   thrust::inclusive_scan                   SS/cache/cache.cu:471-472,500 -> rocPRIM scan
//   GetEdgeMem                               SS/cache/cache_impl.cuh:63-69
//   InitPair / InitIndexPair / InitOffsetPair SS/cache/cache_impl.cuh:89-109 (+ BGHT insert) ->
//       direct-mapped int32/int8 tables indexed by vertex id (N*4 B is trivial in 288 GB HBM)
//   FeatFillUp                               SS/cache/cache_impl.cuh:183-188
//   GetNeighborCount / TopoFillUp            SS/storage/graph_storage_impl.cuh:33-53
#include "legion_core.h"

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/functional.hpp>

namespace lg {

static inline int32_t grid_for(int64_t n, int32_t block = 256, int32_t cap = 4096)
{
    int64_t g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int32_t)g;
}

__global__ void aggregate_access_kernel(unsigned long long* __restrict__ agg,
                                        const unsigned long long* __restrict__ add, int32_t n)
{
    for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        agg[i] += add[i];
}

void aggregate_access(hipStream_t s, unsigned long long* agg, const unsigned long long* add, int32_t n)
{
    aggregate_access_kernel<<<grid_for(n), 256, 0, s>>>(agg, add, n);
    hipCheckError();
}

__global__ void iota_kernel(int32_t* __restrict__ p, int32_t n)
{
    for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = i;
}

// keys_inout: aggregated hotness in vertex order on entry, sorted descending on exit.
// order_out : vertex ids in that order; ties keep ascending id (LSD radix sort is stable).
void sort_hotness_desc(hipStream_t s, unsigned long long* keys_inout, int32_t* order_out, int32_t n)
{
    unsigned long long* keys_tmp = nullptr;
    int32_t* iota = nullptr;
    HIP_CALL(hipMalloc(&keys_tmp, (size_t)n * sizeof(unsigned long long)));
    HIP_CALL(hipMalloc(&iota, (size_t)n * sizeof(int32_t)));
    iota_kernel<<<grid_for(n), 256, 0, s>>>(iota, n);
    hipCheckError();
    size_t temp_bytes = 0;
    HIP_CALL(rocprim::radix_sort_pairs_desc(nullptr, temp_bytes, keys_inout, keys_tmp, iota, order_out,
                                            (size_t)n, 0, 64, s));
    void* temp = nullptr;
    HIP_CALL(hipMalloc(&temp, temp_bytes ? temp_bytes : 16));
    HIP_CALL(rocprim::radix_sort_pairs_desc(temp, temp_bytes, keys_inout, keys_tmp, iota, order_out,
                                            (size_t)n, 0, 64, s));
    HIP_CALL(hipMemcpyAsync(keys_inout, keys_tmp, (size_t)n * sizeof(unsigned long long),
                            hipMemcpyDeviceToDevice, s));
    HIP_CALL(hipStreamSynchronize(s));
    HIP_CALL(hipFree(temp));
    HIP_CALL(hipFree(keys_tmp));
    HIP_CALL(hipFree(iota));
}

template <typename T>
static void inclusive_scan_impl(hipStream_t s, const T* in, T* out, int32_t n)
{
    size_t temp_bytes = 0;
    HIP_CALL(rocprim::inclusive_scan(nullptr, temp_bytes, in, out, (size_t)n, rocprim::plus<T>(), s));
    void* temp = nullptr;
    HIP_CALL(hipMalloc(&temp, temp_bytes ? temp_bytes : 16));
    HIP_CALL(rocprim::inclusive_scan(temp, temp_bytes, in, out, (size_t)n, rocprim::plus<T>(), s));
    HIP_CALL(hipStreamSynchronize(s));
    HIP_CALL(hipFree(temp));
}

void inclusive_scan_u64(hipStream_t s, const unsigned long long* in, unsigned long long* out, int32_t n)
{
    inclusive_scan_impl<unsigned long long>(s, in, out, n);
}

void inclusive_scan_i64(hipStream_t s, const int64_t* in, int64_t* out, int32_t n)
{
    inclusive_scan_impl<int64_t>(s, in, out, n);
}

__global__ void edge_mem_kernel(const int32_t* __restrict__ order, unsigned long long* __restrict__ edge_mem,
                                int32_t n, const int64_t* __restrict__ csr_index)
{
    for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int32_t id = order[i];
        const int64_t neighbor_count = csr_index[id + 1] - csr_index[id];
        edge_mem[i] = sizeof(int64_t) + sizeof(int32_t) * neighbor_count;
    }
}

void edge_mem_in_order(hipStream_t s, const int32_t* order, unsigned long long* edge_mem, int32_t n,
                       const int64_t* csr_index)
{
    edge_mem_kernel<<<grid_for(n), 256, 0, s>>>(order, edge_mem, n, csr_index);
    hipCheckError();
}

template <typename T>
__global__ void fill_kernel(T* __restrict__ p, T v, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        p[i] = v;
}

void fill_value_i32(hipStream_t s, int32_t* p, int32_t v, int64_t n)
{
    fill_kernel<int32_t><<<grid_for(n), 256, 0, s>>>(p, v, n);
    hipCheckError();
}

void fill_value_i8(hipStream_t s, char* p, char v, int64_t n)
{
    fill_kernel<char><<<grid_for(n), 256, 0, s>>>(p, v, n);
    hipCheckError();
}

// node_map[QF[t]] = (t % Kg) * capacity + t / Kg for t < capacity*Kg   (InitPair, cache_impl.cuh:104-109)
__global__ void init_node_map_kernel(int32_t* __restrict__ node_map, const int32_t* __restrict__ QF,
                                     int32_t capacity, int32_t Kg, int32_t n)
{
    const int64_t total = min((int64_t)capacity * Kg, (int64_t)n);   // QF has n entries
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x)
        node_map[QF[t]] = (int32_t)((t % Kg) * capacity + t / Kg);
}

// column slots (legion_core.h, GraphStorage): pair every column entry with the feature-cache slot of the neighbour it names
__global__ void build_column_slots_kernel(const int32_t* __restrict__ col, const int32_t* __restrict__ node_map,
                                          int32_t* __restrict__ pairs, int64_t num_edges)
{
    typedef int32_t v2i __attribute__((ext_vector_type(2)));
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < num_edges; e += (int64_t)gridDim.x * blockDim.x) {
        const int32_t id = col[e];
        v2i v;
        v.x = id;
        v.y = id >= 0 ? node_map[id] : CACHEMISS_FLAG;
        ((v2i*)pairs)[e] = v;
    }
}

void build_column_slots(hipStream_t s, const int32_t* col, const int32_t* node_map, int32_t* colx_pairs, int64_t num_edges)
{
    if (num_edges <= 0) return;
    build_column_slots_kernel<<<8192, 256, 0, s>>>(col, node_map, colx_pairs, num_edges);
    hipCheckError();
}

void init_node_map(hipStream_t s, int32_t* node_map, const int32_t* QF, int32_t capacity, int32_t Kg,
                   int32_t n)
{
    init_node_map_kernel<<<grid_for((int64_t)capacity * Kg), 256, 0, s>>>(node_map, QF, capacity, Kg, n);
    hipCheckError();
}

// HybridInitPair, cache_impl.cuh:113-123: the gpu_cap hottest ids -> slots cpu_cap + t (the GPU cache), the next cpu_cap ids ->
// slots t - gpu_cap (the CPU cache); QF has n entries
__global__ void init_node_map_hybrid_kernel(int32_t* __restrict__ node_map, const int32_t* __restrict__ QF,
                                            int32_t cpu_cap, int32_t gpu_cap, int32_t n)
{
    const int64_t total = min((int64_t)cpu_cap + gpu_cap, (int64_t)n);
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x)
        node_map[QF[t]] = (int32_t)(t < gpu_cap ? cpu_cap + t : t - gpu_cap);
}

void init_node_map_hybrid(hipStream_t s, int32_t* node_map, const int32_t* QF, int32_t cpu_cache_capacity,
                          int32_t gpu_cache_capacity, int32_t n)
{
    init_node_map_hybrid_kernel<<<grid_for((int64_t)cpu_cache_capacity + gpu_cache_capacity), 256, 0, s>>>(
        node_map, QF, cpu_cache_capacity, gpu_cache_capacity, n);
    hipCheckError();
}

// InitIndexPair / InitOffsetPair, cache_impl.cuh:89-101
__global__ void init_edge_maps_kernel(char* __restrict__ index_map, int32_t* __restrict__ offset_map,
                                      const int32_t* __restrict__ QT, int32_t capacity, int32_t Kg,
                                      int32_t Ki, int32_t n)
{
    const int64_t total = min((int64_t)capacity * Kg, (int64_t)n);
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int32_t id = QT[t];
        index_map[id] = (char)(t % Kg + Ki * Kg);
        offset_map[id] = (int32_t)(t / Kg);
    }
}

void init_edge_maps(hipStream_t s, char* index_map, int32_t* offset_map, const int32_t* QT,
                    int32_t capacity, int32_t Kg, int32_t Ki, int32_t n)
{
    init_edge_maps_kernel<<<grid_for((int64_t)capacity * Kg), 256, 0, s>>>(index_map, offset_map, QT,
                                                                         capacity, Kg, Ki, n);
    hipCheckError();
}

// FeatFillUp, cache_impl.cuh:183-188: one 64-lane wave per cached row, 16 B per lane when D%4==0
__global__ void feat_fill_up_kernel(int32_t capacity, int32_t D, float* __restrict__ cache,
                                    const float* __restrict__ table, const int32_t* __restrict__ QF,
                                    int32_t Kg, int32_t Ki, int32_t n)
{
    const int32_t lane = threadIdx.x & 63;
    const int32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int32_t nwaves = (gridDim.x * blockDim.x) >> 6;
    for (int32_t r = wave; r < capacity; r += nwaves) {
        const int64_t t = (int64_t)r * Kg + Ki;
        if (t >= n) continue;
        const int32_t id = QF[t];
        const float* src = table + (int64_t)id * D;
        float* dst = cache + (int64_t)r * D;
        if ((D & 3) == 0) {
            typedef float v4 __attribute__((ext_vector_type(4)));
            for (int32_t c = lane; c < D / 4; c += 64)
                reinterpret_cast<v4*>(dst)[c] = reinterpret_cast<const v4*>(src)[c];
        } else {
            for (int32_t c = lane; c < D; c += 64) dst[c] = src[c];
        }
    }
}

void feat_fill_up(hipStream_t s, int32_t capacity, int32_t D, float* cache, const float* table,
                  const int32_t* QF, int32_t Kg, int32_t Ki, int32_t n)
{
    if (capacity <= 0 || D <= 0) return;
    feat_fill_up_kernel<<<grid_for((int64_t)capacity * 64, 256, 8192), 256, 0, s>>>(capacity, D, cache, table, QF, Kg, Ki, n);
    hipCheckError();
}

__global__ void topo_neighbor_count_kernel(const int32_t* __restrict__ QT, int32_t Kg, int32_t Ki,
                                           int32_t capacity, int32_t n,
                                           const int64_t* __restrict__ csr_index,
                                           int64_t* __restrict__ counts)
{
    for (int32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < capacity; r += gridDim.x * blockDim.x) {
        const int64_t t = (int64_t)r * Kg + Ki;
        int64_t c = 0;
        if (t < n) {
            const int32_t id = QT[t];
            c = csr_index[id + 1] - csr_index[id];
        }
        counts[r] = c;
    }
}

void topo_neighbor_count(hipStream_t s, const int32_t* QT, int32_t Kg, int32_t Ki, int32_t capacity,
                         int32_t n, const int64_t* csr_index, int64_t* counts)
{
    topo_neighbor_count_kernel<<<grid_for(capacity), 256, 0, s>>>(QT, Kg, Ki, capacity, n, csr_index, counts);
    hipCheckError();
}

// TopoFillUp: the reference walks a whole adjacency per thread; here one wave per vertex
__global__ void topo_fill_up_kernel(const int32_t* __restrict__ QT, int32_t Kg, int32_t Ki,
                                    int32_t capacity, int32_t n, const int64_t* __restrict__ csr_index,
                                    const int32_t* __restrict__ csr_dst,
                                    const int64_t* __restrict__ d_index, int32_t* __restrict__ d_dst)
{
    const int32_t lane = threadIdx.x & 63;
    const int32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int32_t nwaves = (gridDim.x * blockDim.x) >> 6;
    for (int32_t r = wave; r < capacity; r += nwaves) {
        const int64_t t = (int64_t)r * Kg + Ki;
        if (t >= n) continue;
        const int32_t id = QT[t];
        const int64_t begin = csr_index[id], cnt = csr_index[id + 1] - begin, out = d_index[r];
        for (int64_t i = lane; i < cnt; i += 64) d_dst[out + i] = csr_dst[begin + i];
    }
}

void topo_fill_up(hipStream_t s, const int32_t* QT, int32_t Kg, int32_t Ki, int32_t capacity, int32_t n,
                  const int64_t* csr_index, const int32_t* csr_dst, const int64_t* d_index, int32_t* d_dst)
{
    if (capacity <= 0) return;
    topo_fill_up_kernel<<<grid_for((int64_t)capacity * 64, 256, 8192), 256, 0, s>>>(QT, Kg, Ki, capacity, n,
                                                                                   csr_index, csr_dst, d_index, d_dst);
    hipCheckError();
}

}  // namespace lg
