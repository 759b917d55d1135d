// kernels_gather.hip -- feature-cache lookup + row gather for gfx950.
//
// Computes what multiGPU_feat_cache_lookup does (SS/cache/cache_impl.cuh:239-272, launched from
// SS/cache/cache.cu:726-748) with PreSCCacheController::FindFeat (SS/cache/cache.cu:180-215)
// fused in front of it: for row r of the current new-node range
//     id = sampled_ids[off + r];  g = node_map[id] (or -2)          -> cache_index[r] = g
//     g <  0 : dst[off + r] = full_table[id % N]                    (miss: full table tier)
//     g >= 0 : dst[off + r] = cache_tables[g / cap][g % cap]        (hit: local or peer HBM)
// Rows are copied verbatim (byte-identical, no arithmetic).
//
// The reference runs one thread per float on a fixed 32x1024 grid with a 64-bit div/mod per
// element.  Here a workgroup owns a tile of 64 consecutive output rows: 64 lanes resolve the
// tile's ids to source row pointers once (the id->slot table is direct mapped, one 4-byte read)
// and park them in LDS; then all 256 lanes stream the tile as 16-byte chunks, four independent
// loads in flight per lane, so that consecutive lanes read consecutive 16 B of one source row
// and write consecutive 16 B of the (contiguous) destination.  No divides in the copy loop.
//
// Roofline: HBM.  Algorithmic bytes per row = 8*D + 8 (D*4 read + D*4 written + id + index).
#include "legion_core.h"

namespace lg {

#define LG_GATHER_ROWS 64
#define LG_GATHER_THREADS 256
#define LG_GATHER_UNROLL 4

template <typename VecT>
__global__ __launch_bounds__(LG_GATHER_THREADS) void gather_kernel(
    const float* __restrict__ full_table, const float* const* __restrict__ cache_tables,
    const int32_t* __restrict__ node_map, int32_t node_capacity, int32_t D, int32_t total_num_nodes,
    const int32_t* __restrict__ sampled_ids, int32_t* __restrict__ cache_index_out,
    const int32_t* __restrict__ range, int32_t* __restrict__ range_copy, float* __restrict__ dst,
    int32_t max_rows, int32_t dst_rows)
{
    constexpr int VEC = sizeof(VecT) / sizeof(float);
    __shared__ const float* s_ptr[LG_GATHER_ROWS];

    const int32_t off = range[0];
    int32_t rows = range[1];
    if (blockIdx.x == 0 && threadIdx.x == 0 && range_copy) {   // counter_update(op%3==1), operator_impl.cu:83-85
        range_copy[0] = off;
        range_copy[1] = rows;
    }
    if (rows > max_rows) rows = max_rows;
    if (rows > dst_rows - off) rows = dst_rows - off;   // never write past the feature buffer (the
                                                        // reference sizes it 1.2 x PreSC max and would overrun)
    const int32_t ntiles = (rows + LG_GATHER_ROWS - 1) / LG_GATHER_ROWS;
    const int32_t tid = threadIdx.x;
    const int32_t C = D / VEC;                         // chunks per row
    const int32_t dr = LG_GATHER_THREADS / C;          // row / chunk advance per 256-chunk step
    const int32_t dc = LG_GATHER_THREADS - dr * C;

    // one tile per workgroup: the grid covers the whole feature buffer, surplus workgroups leave
    // here, and the hardware dispatcher balances the rest (a grid-stride loop over a fixed grid
    // left a 1-vs-2-tiles imbalance at typical sizes)
    {
        const int32_t tile = blockIdx.x;
        if (tile >= ntiles) return;
        const int32_t r0 = tile * LG_GATHER_ROWS;
        const int32_t nr = min(LG_GATHER_ROWS, rows - r0);
        if (tid < nr) {
            const int32_t id = sampled_ids[off + r0 + tid];
            int32_t g = CACHEMISS_FLAG;
            if (node_map != nullptr && id >= 0) g = node_map[id];
            cache_index_out[r0 + tid] = g;             // FindFeat writes from index 0 each hop
            const float* p = nullptr;
            if (g < 0) {
                if (id >= 0) p = full_table + (int64_t)(id % total_num_nodes) * D;   // :262-266
            } else {
                const int32_t didx = g / node_capacity, fidx = g - didx * node_capacity;   // :259-260
                p = cache_tables[didx] + (int64_t)fidx * D;                                  // :268
            }
            s_ptr[tid] = p;
        }
        __syncthreads();

        const int32_t nchunks = nr * C;
        float* dst_tile = dst + (int64_t)(off + r0) * D;
        int32_t q = tid;
        int32_t r = q / C;
        int32_t c = q - r * C;
        while (q < nchunks) {
            VecT v[LG_GATHER_UNROLL];
            int32_t rr[LG_GATHER_UNROLL], cc[LG_GATHER_UNROLL];
            bool ok[LG_GATHER_UNROLL];
#pragma unroll
            for (int u = 0; u < LG_GATHER_UNROLL; u++) {
                rr[u] = r;
                cc[u] = c;
                ok[u] = false;
                if (q < nchunks) {
                    const float* p = s_ptr[r];
                    if (p != nullptr) {
                        v[u] = __builtin_nontemporal_load(reinterpret_cast<const VecT*>(p) + c);
                        ok[u] = true;
                    }
                }
                q += LG_GATHER_THREADS;
                r += dr;
                c += dc;
                if (c >= C) { c -= C; r += 1; }
            }
#pragma unroll
            for (int u = 0; u < LG_GATHER_UNROLL; u++) {
                if (ok[u])
                    __builtin_nontemporal_store(v[u], reinterpret_cast<VecT*>(dst_tile + (int64_t)rr[u] * D) + cc[u]);
            }
        }
    }
}

void launch_gather(hipStream_t s, const float* full_table, const float* const* cache_tables,
                   const int32_t* node_map, int32_t node_capacity, int32_t D, int32_t total_num_nodes,
                   const int32_t* sampled_ids, int32_t* cache_index_out, const int32_t* range,
                   int32_t* range_copy, float* dst, int32_t max_rows, int32_t dst_rows)
{
    if (D <= 0 || max_rows <= 0) return;                // :256 float_feature_len > 0
    if (node_capacity < 1) node_capacity = 1;
    const int32_t grid = (max_rows + LG_GATHER_ROWS - 1) / LG_GATHER_ROWS;
    typedef float v4 __attribute__((ext_vector_type(4)));
    typedef float v2 __attribute__((ext_vector_type(2)));
    if (D % 4 == 0)
        gather_kernel<v4><<<grid, LG_GATHER_THREADS, 0, s>>>(full_table, cache_tables, node_map,
            node_capacity, D, total_num_nodes, sampled_ids, cache_index_out, range, range_copy, dst, max_rows, dst_rows);
    else if (D % 2 == 0)
        gather_kernel<v2><<<grid, LG_GATHER_THREADS, 0, s>>>(full_table, cache_tables, node_map,
            node_capacity, D, total_num_nodes, sampled_ids, cache_index_out, range, range_copy, dst, max_rows, dst_rows);
    else
        gather_kernel<float><<<grid, LG_GATHER_THREADS, 0, s>>>(full_table, cache_tables, node_map,
            node_capacity, D, total_num_nodes, sampled_ids, cache_index_out, range, range_copy, dst, max_rows, dst_rows);
    hipCheckError();
}

}  // namespace lg
