// kernels_gather.hip -- feature-cache lookup + row gather for gfx950.
//
// Computes what multiGPU_feat_cache_lookup does (SS/cache/cache_impl.cuh:239-272, launched from
// SS/cache/cache.cu:726-748) with PreSCCacheController::FindFeat (SS/cache/cache.cu:180-215)
// fused in front of it: for row r of the current new-node range
//     id = sampled_ids[off + r];  g = node_map[id] (or -2)          -> cache_index[r] = g
//     g <  0 : dst[off + r] = full_table[id % N]                    (miss: full table tier)
//     g >= 0 : dst[off + r] = cache_tables[g / cap][g % cap]        (hit: local or peer HBM)
// and, with GatherParams.hybrid, what feat_cache_lookup does (SS/cache/cache_impl.cuh:202-235):
//     0 <= g < cpu_cap : dst[off + r] = cpu_cache[g]                (hit in the mapped pinned CPU cache)
//     g >= cpu_cap     : dst[off + r] = gpu_cache[(g - cpu_cap) % gpu_cap]
//     g <  0           : left to the storage tier (the full table when one is bound, else the row is not written)
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

#include <cstdlib>
#include <cstring>

namespace lg {

#define LG_GATHER_ROWS 64
#define LG_GATHER_THREADS 256
#define LG_GATHER_UNROLL 4

// ------------------------------------------------------------------------------------------
// Hand-over of a finished mini-batch from a lane of a launch group to one of the two trainer-visible pipe
// slots (SS/engine/ipc_service.cu:134-211: ids, labels, agg_src, agg_dst, node_counter, edge_counter; the
// feature rows are gathered straight into the slot by gather_kernel).  Sizes come from the lane's counters on
// the device.  The counters also go to the slot's host-visible mirror (ipc_env.hip), with node_counter[2..3]
// already holding what the last gather op leaves there (counter_update(op%3==1), operator_impl.cu:83-85).
// ------------------------------------------------------------------------------------------
// slice `part` of `parts` of the hand-over, done by one workgroup (all of it when parts == 1)
__device__ __forceinline__ void deliver_slice(const LanePtrs& L, const DeliverParams& d, int32_t part, int32_t parts)
{
    const LG_G int32_t* nc = LG_GPTR(const int32_t, L.node_counter);
    const LG_G int32_t* ec = LG_GPTR(const int32_t, L.edge_counter);
    const LG_G int32_t* hs = LG_GPTR(const int32_t, L.hop_scratch);
    const int32_t hop_num = nc[INTRABATCH_CON * 3 - 1];
    int32_t n = nc[INTRABATCH_CON * 3 + hop_num], e = ec[INTRABATCH_CON * 3 + hop_num], b = nc[INTRABATCH_CON * 3];
    n = n < 0 ? 0 : (n > d.num_ids ? d.num_ids : n);
    e = e < 0 ? 0 : (e > d.num_ids ? d.num_ids : e);
    b = b < 0 ? 0 : (b > d.batch_cap ? d.batch_cap : b);
    const int32_t tid = threadIdx.x, nthr = blockDim.x;
    typedef int32_t v4 __attribute__((ext_vector_type(4)));
    auto copy = [&](const int32_t* src, int32_t* dst, int32_t count) {
        const int32_t per = (((count + parts - 1) / parts) + 3) & ~3;     // slices start 16-byte aligned
        const int32_t lo = part * per, hi = min(lo + per, count);
        if (lo >= hi) return;
        const LG_G v4* s4 = (const LG_G v4*)(src + lo);
        LG_G v4* d4 = (LG_G v4*)(dst + lo);
        const int32_t q = (hi - lo) >> 2;
        for (int32_t i = tid; i < q; i += nthr) d4[i] = s4[i];
        for (int32_t i = lo + (q << 2) + tid; i < hi; i += nthr) LG_GPTR(int32_t, dst)[i] = LG_GPTR(const int32_t, src)[i];
    };
    copy(L.sampled_ids, d.sampled_ids, n);
    copy(L.agg_src_off, d.agg_src_off, e);
    copy(L.agg_dst_off, d.agg_dst_off, e);
    copy(L.labels, d.labels, b);
    if (part == 0 && tid < 32) {
        int32_t v = tid < 16 ? nc[tid] : ec[tid - 16];
        if (tid == 2) v = hs[HS_RANGE + 2 * hop_num];
        if (tid == 3) v = hs[HS_RANGE + 2 * hop_num + 1];
        LG_GPTR(int32_t, tid < 16 ? d.node_counter : d.edge_counter)[tid & 15] = v;
        if (d.mirror != nullptr) LG_GPTR(int32_t, d.mirror)[tid] = v;      // host memory: visible once the batch's event completed
    }
}

__global__ __launch_bounds__(256) void deliver_kernel(const LanePtrs* __restrict__ lane_p, DeliverParams d)
{
    deliver_slice(*lane_p, d, blockIdx.x, gridDim.x);
}

void launch_deliver(hipStream_t s, const LanePtrs* d_lane, const DeliverParams& d)
{
    int32_t grid = (d.num_ids / 4 + 1023) / 1024;
    if (grid > 512) grid = 512;
    if (grid < 1) grid = 1;
    deliver_kernel<<<grid, 256, 0, s>>>(d_lane, d);
    hipCheckError();
}

// VecT: float4 for rows that are multiples of 16 bytes; `v4u` -- the same 16 bytes per lane at 4-byte alignment -- for every other
// width of at least 4 floats (gfx950 global loads / stores of 16 bytes need dword alignment only; the compiler emits
// global_load_dwordx4 for both), with the D % 4 trailing floats of each row moved by a scalar pass (TAIL); float below that.
//
// A workgroup walks the tiles blockIdx.x, blockIdx.x + gridDim.x, ... of its lane and keeps the walk software-pipelined: while
// tile t is being copied, the first ROWS threads already hold the ids of tile t+1 (loaded one step earlier), have its
// slot lookups (node_map[id], the one scattered read of the resolve) in flight together with the copy's loads, and fetch the
// ids of tile t+2.  The copy of a tile therefore never waits for its own resolve chain (id -> slot -> pointer): with one tile
// per workgroup that chain was 40 % of a workgroup's life during which it streamed nothing.  Loads of the copy loop are
// unconditional (a missing row reads the destination instead, chunks past the tile's end re-read its last chunk), so no
// branch sits between the prefetch and the copy and all of them are in flight together; only the stores are predicated.
#ifndef LG_GATHER_MIN_WAVES
#define LG_GATHER_MIN_WAVES 8        // waves per SIMD the register allocation must leave room for
#endif
#ifndef LG_GATHER_TARGET_WG
#define LG_GATHER_TARGET_WG 8192     // workgroups of a full launch: four rounds of what is resident, so the dispatcher evens out lanes of different length
#endif
// LASTOP: the gather of a batch's last op (the dominant launch; traces and counters tell it from the early hops' gathers by
// name) -- the only one that can carry a hand-over to a trainer-visible pipe slot.
template <typename VecT, int ROWS = LG_GATHER_ROWS, int UNROLL = LG_GATHER_UNROLL, bool TAIL = false, bool LASTOP = true>
__global__ __launch_bounds__(LG_GATHER_THREADS, LG_GATHER_MIN_WAVES) void gather_kernel(GatherParams gp, const LanePtrs* __restrict__ lanes,
                                                                   bool copy_range)
{
    constexpr int VEC = sizeof(VecT) / sizeof(float);
    static_assert(ROWS <= LG_GATHER_THREADS, "one resolving thread per row of a tile");
    __shared__ const LG_G float* s_ptr[2][ROWS];

    const LanePtrs& L = lanes[blockIdx.y];
    // {offset, count} of the new-node range: the live counters, or the per-hop snapshot
    const LG_G int32_t* range = LG_GPTR(const int32_t, gp.hop >= 0 ? L.hop_scratch + HS_RANGE + 2 * gp.hop : L.node_counter);
    const LG_G int32_t* sampled_ids = LG_GPTR(const int32_t, L.sampled_ids);
    const LG_G int32_t* node_slot = (L.node_slot != nullptr && gp.node_map != nullptr) ? LG_GPTR(const int32_t, L.node_slot) : nullptr;
    LG_G int32_t* cache_search_buffer = LG_GPTR(int32_t, L.cache_search_buffer);
    int32_t off = range[0];
    int32_t rows = range[1];
    if (copy_range && blockIdx.x == 0 && threadIdx.x == 0) {   // counter_update(op%3==1), operator_impl.cu:83-85
        LG_GPTR(int32_t, L.node_counter)[2] = off;
        LG_GPTR(int32_t, L.node_counter)[3] = rows;
    }
    if (gp.hop >= 0 && gp.first_hop < gp.hop) {                // earlier hops' ranges ride along (adjacent rows)
        const int32_t off0 = LG_GPTR(const int32_t, L.hop_scratch)[HS_RANGE + 2 * gp.first_hop];
        rows += off - off0;
        off = off0;
    }
    if (rows > gp.max_rows) rows = gp.max_rows;
    if (rows > L.feature_rows - off) {     // never write past the feature buffer (the reference sizes it 1.2 x the PreSC
        rows = L.feature_rows - off;       // maximum and would overrun, SS/engine/server.cu:277): stop at its end and say so
        if (blockIdx.x == 0 && threadIdx.x == 0 && L.hop_scratch != nullptr) {
            __hip_atomic_fetch_or(LG_GPTR(int32_t, L.hop_scratch) + HS_ERROR, LG_ERR_FEATURE_ROWS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (L.err_flag) __hip_atomic_fetch_or(LG_GPTR(int32_t, L.err_flag), LG_ERR_FEATURE_ROWS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    const int32_t ntiles = (rows + ROWS - 1) / ROWS;
    const int32_t step = gridDim.x;
    if (LASTOP && L.deliver != nullptr)        // this lane's gather also hands the batch over to a pipe slot: every workgroup a slice
        deliver_slice(L, *static_cast<const DeliverParams*>(L.deliver), blockIdx.x, step);
    const int32_t tid = threadIdx.x;
    const int32_t D = gp.D;
    const int32_t C = D / VEC;                         // chunks per row
    const int32_t dr = LG_GATHER_THREADS / C;          // row / chunk advance per 256-chunk step
    const int32_t dc = LG_GATHER_THREADS - dr * C;
    int32_t tile = blockIdx.x;
    if (tile >= ntiles) return;                        // (surplus workgroups of a short lane)

    // row-source statistics: armed by the host (UnifiedCache::GatherStats) and switched by a DEVICE word, so that a launch
    // captured in a hipGraph follows the switch too (a pointer baked in at capture time kept counting through every replay)
    const bool counting = gp.stats != nullptr && gp.Kg > 1 && gp.stats[3] != 0ull;
    // a row's source: FindFeat (cache.cu:180-215) + the address arithmetic of cache_impl.cuh:259-268
    auto source_of = [&](int32_t id, int32_t g) -> const LG_G float* {
        const LG_G float* p = nullptr;
        if (g < 0) {
            if (id >= 0 && gp.full_table != nullptr)     // :262-266 (the modulo only where it does anything)
                p = LG_GPTR(const float, gp.full_table) + (int64_t)(id < gp.total_num_nodes ? id : id % gp.total_num_nodes) * D;
        } else if (gp.hybrid) {      // feat_cache_lookup, cache_impl.cuh:224-231: CPU cache below cpu_cap, this GPU's cache above
            if (g < gp.hybrid_cpu_cap)
                p = LG_GPTR(const float, gp.hybrid_cpu_cache) + (int64_t)g * D;       // (g % cpu_cap == g)
            else
                p = LG_GPTR(const float, gp.local_table) + (int64_t)((g - gp.hybrid_cpu_cap) % gp.hybrid_gpu_cap) * D;
        } else {
            int32_t didx = 0, fidx = g;                                                      // :259-260 (one division, and
            if (gp.striped) { didx = g / gp.node_capacity; fidx = g - didx * gp.node_capacity; }  // none without striping)
            const int64_t rank = (int64_t)fidx * gp.Kg + didx;                               // hotness rank of the row (cache_impl.cuh:104-109)
            const bool local_copy = gp.replica != nullptr && rank < gp.replica_rows;
            if (local_copy)      // the clique's hottest rows are also kept locally: same row, no xGMI hop
                p = LG_GPTR(const float, gp.replica) + rank * gp.D;
            else if (didx == gp.member && gp.local_table != nullptr)     // own stripe: its address came with the launch
                p = LG_GPTR(const float, gp.local_table) + (int64_t)fidx * gp.D;
            else if (gp.skip_remote)     // peer_gather = bulk: the owner pushes this row (bulk_push_kernel); nothing to fetch here
                p = nullptr;
            else
                p = LG_GPTR(const float, gp.cache_tables[didx]) + (int64_t)fidx * gp.D;                  // :268
            if (counting) {    // tests / diagnostics / the computed xGMI count: [0] rows read through a stripe pointer, [1] from the
                               // replica, [2] the part of [0] from a peer's stripe -- one atomic per wave and counter
                const unsigned long long m_rep = __ballot(local_copy), m_str = __ballot(!local_copy);
                const unsigned long long m_peer = __ballot(!local_copy && didx != gp.member);
                if ((int)(threadIdx.x & 63) == __ffsll((unsigned long long)(m_rep | m_str)) - 1) {
                    if (m_str) atomicAdd(gp.stats + 0, (unsigned long long)__popcll(m_str));
                    if (m_rep) atomicAdd(gp.stats + 1, (unsigned long long)__popcll(m_rep));
                    if (m_peer) atomicAdd(gp.stats + 2, (unsigned long long)__popcll(m_peer));
                }
            }
        }
        return p;
    };
    // the row's feature-cache slot: carried from the sampler where the column slots are in use (a coalesced read), else --
    // seeds, rows sampled from a cached-topology CSR, no column slots -- the FindFeat lookup (a 128-byte line per row)
    auto lookup = [&](int32_t id, int32_t carried) -> int32_t {
        if (carried != LG_FS_UNKNOWN) return carried;
        return (gp.node_map != nullptr && id >= 0) ? gp.node_map[id] : CACHEMISS_FLAG;
    };

    // prologue: this workgroup's first tile resolved, the second tile's ids in hand
    int32_t id_n = -1, fs_n = LG_FS_UNKNOWN;           // ids / carried slots of the NEXT tile (threads < ROWS)
    bool have_n = false;
    if (tid < ROWS) {
        const int32_t r = tile * ROWS + tid;
        if (r < rows) {
            const int32_t id = sampled_ids[off + r];
            const int32_t g = lookup(id, node_slot != nullptr ? node_slot[off + r] : LG_FS_UNKNOWN);
            cache_search_buffer[r] = g;                // FindFeat writes from index 0 each hop
            s_ptr[0][tid] = source_of(id, g);
        }
        const int32_t rn = (tile + step) * ROWS + tid;
        have_n = tile + step < ntiles && rn < rows;
        if (have_n) {
            id_n = sampled_ids[off + rn];
            if (node_slot != nullptr) fs_n = node_slot[off + rn];
        }
    }
    __syncthreads();

    int buf = 0;
    for (; tile < ntiles; tile += step) {
        const int32_t r0 = tile * ROWS;
        const int32_t nr = min(ROWS, rows - r0);
        // resolve of the next tile: its one scattered read goes out before the copy's loads; the ids of the tile after it too
        int32_t g_n = CACHEMISS_FLAG, id_nn = -1, fs_nn = LG_FS_UNKNOWN;
        bool have_nn = false;
        if (tid < ROWS) {
            if (have_n) g_n = lookup(id_n, fs_n);
            const int32_t rnn = (tile + 2 * step) * ROWS + tid;
            have_nn = tile + 2 * step < ntiles && rnn < rows;
            if (have_nn) {
                id_nn = sampled_ids[off + rnn];
                if (node_slot != nullptr) fs_nn = node_slot[off + rnn];
            }
        }

        const int32_t nchunks = nr * C;
        LG_G float* dst_tile = LG_GPTR(float, L.float_features) + (int64_t)(off + r0) * D;
        int32_t q = tid;
        int32_t r = q / C;
        int32_t c = q - r * C;
        while (q < nchunks) {
            VecT v[UNROLL];
            int32_t rr[UNROLL], cc[UNROLL];
            bool ok[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; u++) {
                const bool in = q < nchunks;
                rr[u] = in ? r : nr - 1;               // past the end: the tile's last chunk again (loaded, not stored)
                cc[u] = in ? c : C - 1;
                const LG_G float* p = s_ptr[buf][rr[u]];
                ok[u] = in && p != nullptr;
                if (p == nullptr) p = dst_tile + (int64_t)rr[u] * D;     // id < 0: nothing to fetch; read what is there
                v[u] = ((const LG_G VecT*)p)[cc[u]];   // plain loads: measured 74% of HBM peak vs 63% nontemporal
                q += LG_GATHER_THREADS;
                r += dr;
                c += dc;
                if (c >= C) { c -= C; r += 1; }
            }
#pragma unroll
            for (int u = 0; u < UNROLL; u++) {
                if (ok[u])     // write-once output: nontemporal stores
                    __builtin_nontemporal_store(v[u], (LG_G VecT*)(dst_tile + (int64_t)rr[u] * D) + cc[u]);
            }
        }
        if (TAIL) {            // the last D % VEC floats of every row
            const int32_t tail = D - C * VEC;
            for (int32_t i = tid; i < nr * tail; i += LG_GATHER_THREADS) {
                const int32_t tr = i / tail, k = C * VEC + (i - tr * tail);
                const LG_G float* p = s_ptr[buf][tr];
                if (p != nullptr) dst_tile[(int64_t)tr * D + k] = p[k];
            }
        }
        if (tid < ROWS && have_n) {
            cache_search_buffer[(tile + step) * ROWS + tid] = g_n;
            s_ptr[buf ^ 1][tid] = source_of(id_n, g_n);
        }
        id_n = id_nn;
        fs_n = fs_nn;
        have_n = have_nn;
        __syncthreads();
        buf ^= 1;
    }
}

// ------------------------------------------------------------------------------------------
// peer_gather = bulk, requester side: every row of the group's batches that is a hit in ANOTHER member's stripe (and not in the
// local replica) is listed for its owner -- {row inside the owner's stripe, byte offset of the destination row inside this
// GPU's lane arena} -- one reservation per wave and owner (ballot + popcount).  The order inside a list is arbitrary; what lands
// where is not.  Same lookup as the gather's (FindFeat: carried slot or node_map[id]).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bulk_bucket_kernel(GatherParams gp, const LanePtrs* __restrict__ lanes, BulkLists b, const char* arena_base)
{
    const LanePtrs& L = lanes[blockIdx.y];
    const LG_G int32_t* nc = LG_GPTR(const int32_t, L.node_counter);
    const int32_t hop_num = nc[INTRABATCH_CON * 3 - 1];
    int32_t n = nc[INTRABATCH_CON * 3 + hop_num];
    if (n > L.feature_rows) n = L.feature_rows;
    if (n > gp.max_rows) n = gp.max_rows;
    const LG_G int32_t* sampled_ids = LG_GPTR(const int32_t, L.sampled_ids);
    const LG_G int32_t* node_slot = (L.node_slot != nullptr && gp.node_map != nullptr) ? LG_GPTR(const int32_t, L.node_slot) : nullptr;
    const int32_t lane = threadIdx.x & 63;
    for (int32_t r0 = (blockIdx.x * 256 + (threadIdx.x & ~63)); r0 < n; r0 += gridDim.x * 256) {     // a wave takes 64 consecutive rows
        const int32_t r = r0 + lane;
        int32_t owner = -1, fidx = 0;
        if (r < n) {
            const int32_t id = sampled_ids[r];
            int32_t g = node_slot != nullptr ? node_slot[r] : LG_FS_UNKNOWN;
            if (g == LG_FS_UNKNOWN) g = (gp.node_map != nullptr && id >= 0) ? gp.node_map[id] : CACHEMISS_FLAG;
            if (g >= 0) {
                const int32_t didx = g / gp.node_capacity;
                fidx = g - didx * gp.node_capacity;
                const int64_t rank = (int64_t)fidx * gp.Kg + didx;
                const bool local_copy = gp.replica != nullptr && rank < gp.replica_rows;
                if (!local_copy && didx != gp.member) owner = didx;
            }
        }
        for (int32_t o = 0; o < b.Kg; o++) {
            const unsigned long long m = __ballot(owner == o);
            if (m == 0ull) continue;
            unsigned long long base = 0;
            if (lane == __ffsll((long long)m) - 1) base = atomicAdd(b.cnt + o, (unsigned long long)__popcll(m));
            base = __shfl(base, __ffsll((long long)m) - 1);
            if (owner == o) {
                const int64_t at = (int64_t)base + __popcll(m & ((lane == 0) ? 0ull : (~0ull >> (64 - lane))));
                if (at < b.cap) {
                    b.fidx[(int64_t)o * b.cap + at] = fidx;
                    b.dst[(int64_t)o * b.cap + at] = (const char*)(L.float_features + (int64_t)r * gp.D) - arena_base;
                }
            }
        }
    }
}

void launch_bulk_bucket(hipStream_t s, const GatherParams& g, const LanePtrs* d_lanes, int32_t n_lanes, const BulkLists& lists,
                        const char* arena_base)
{
    if (g.D <= 0 || g.max_rows <= 0 || g.node_map == nullptr || g.Kg <= 1) return;
    GatherParams gp = g;
    if (gp.node_capacity < 1) gp.node_capacity = 1;
    int32_t gx = (g.max_rows + 255) / 256;
    while (gx > 16 && (int64_t)gx * n_lanes > 16384) gx = (gx + 1) / 2;
    bulk_bucket_kernel<<<dim3(gx, n_lanes), 256, 0, s>>>(gp, d_lanes, lists, arena_base);
    hipCheckError();
}

// ------------------------------------------------------------------------------------------
// peer_gather = bulk, owner side: the rows a requester listed for THIS GPU's stripe, read from local HBM and written straight to
// their destination rows in the requester's lane arena -- whole rows, 16-byte chunks, consecutive lanes on consecutive chunks:
// over xGMI these are posted stores of contiguous 512-1024-byte runs instead of the requester's scattered load round trips.
// The lists (12 bytes per row) are read through the requester's peer mapping.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bulk_push_kernel(const float* __restrict__ stripe, int32_t D, const int32_t* __restrict__ fidx,
                                                       const int64_t* __restrict__ dst, const unsigned long long* __restrict__ cnt,
                                                       int64_t cap, char* __restrict__ peer_arena)
{
    typedef float v4u __attribute__((ext_vector_type(4), aligned(4)));
    int64_t n = (int64_t)cnt[0];
    if (n > cap) n = cap;
    const int32_t lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * 256) >> 6;
    const int32_t C = D / 4;
    for (int64_t i = wave; i < n; i += nwaves) {
        const LG_G float* src = LG_GPTR(const float, stripe) + (int64_t)fidx[i] * D;
        LG_G float* out = (LG_G float*)(peer_arena + dst[i]);
        for (int32_t c = lane; c < C; c += 64) ((LG_G v4u*)out)[c] = ((const LG_G v4u*)src)[c];
        for (int32_t k = C * 4 + lane; k < D; k += 64) out[k] = src[k];
    }
}

void launch_bulk_push(hipStream_t s, const float* stripe, int32_t D, const int32_t* fidx, const int64_t* dst,
                      const unsigned long long* cnt, int64_t cap, char* peer_arena)
{
    if (stripe == nullptr || D <= 0 || cap <= 0) return;
    bulk_push_kernel<<<2048, 256, 0, s>>>(stripe, D, fidx, dst, cnt, cap, peer_arena);
    hipCheckError();
}

// tiles of a lane are walked by gx workgroups: about LG_GATHER_TARGET_WG workgroups per launch, never more than tiles
static inline int32_t gather_grid_x(int32_t max_rows, int32_t rows_per_tile, int32_t n_lanes)
{
    const int32_t tiles = (max_rows + rows_per_tile - 1) / rows_per_tile;
    int32_t gx = (LG_GATHER_TARGET_WG + n_lanes - 1) / n_lanes;
    if (gx > tiles) gx = tiles;
    return gx < 1 ? 1 : gx;
}

template <int ROWS>
static void launch_gather_v4(hipStream_t s, const GatherParams& g, int32_t grid_rows, const LanePtrs* d_lanes, int32_t n_lanes, bool copy_range)
{
    typedef float v4 __attribute__((ext_vector_type(4)));
    const dim3 grid(gather_grid_x(grid_rows, ROWS, n_lanes), n_lanes);
    if (g.last_op) gather_kernel<v4, ROWS, LG_GATHER_UNROLL, false, true><<<grid, LG_GATHER_THREADS, 0, s>>>(g, d_lanes, copy_range);
    else gather_kernel<v4, ROWS, LG_GATHER_UNROLL, false, false><<<grid, LG_GATHER_THREADS, 0, s>>>(g, d_lanes, copy_range);
}

static void launch_gather_impl(hipStream_t s, GatherParams g_in, const LanePtrs* d_lanes, int32_t n_lanes, bool copy_range)
{
    if (g_in.D <= 0 || g_in.max_rows <= 0) return;      // :256 float_feature_len > 0
    if (g_in.node_capacity < 1) g_in.node_capacity = 1;
    const GatherParams& gk = g_in;                      // what the kernel gets: max_rows = the clamp
    GatherParams g = g_in;                              // what sizes the launch: the rows a lane typically has (GatherParams.grid_rows)
    if (g.grid_rows > 0 && g.grid_rows < g.max_rows) g.max_rows = g.grid_rows;
    const dim3 grid(gather_grid_x(g.max_rows, LG_GATHER_ROWS, n_lanes), n_lanes);     // (the 4-byte vector path, and 64-row tiles at dword alignment)
    const LegionTuning& tune = tuning();
    if (g.D % 4 == 0) {
        // rows per workgroup (LegionTuning.gather_rows_per_wg; 0 = the default below).  A launch of one or a few lanes (the
        // Runner's per-batch hand-over) has too few 64-row tiles to keep 256 CUs busy: 16-row tiles give it 4 x the
        // workgroups; a full lane group is indifferent to the tile size at D = 128
        // Default for a full group, early in round 3 (one tile per workgroup, 256-lane groups, profiles/r03/gather_experiments.txt): the
        // tile whose payload is 32 KB -- D = 256 with 32 rows 0.758 of peak against 0.727 with 64; D = 64 with 128 rows 0.710 against
        // 0.685 with 64; D = 128 with 64 or 128 rows 0.776 / 0.777, with 32 rows 0.751.  Re-measured at the end of the round: below
        int rows = tune.gather_rows_per_wg;
        if (rows <= 0) {
            rows = 16;
            // (round 3, with the pipelined walk and 512-lane groups: 16 KB of payload for rows of 512 bytes and more -- 32 rows at
            // D = 128: the same at the headline, +1.4 % at B = 8000, +2...3 % on the cold three-hop shapes; D = 256 with 16 rows
            // +1 % -- and 32 KB below that: D = 64 with 128 rows 8.31 G edges/s, with 64 rows 8.17 G)
            const int64_t payload = g.D * 4 >= 512 ? 16384 : 32768;
            while (rows < 256 && (int64_t)rows * 2 * g.D * 4 <= payload + payload / 4) rows *= 2;       // D = 100 -> 64, D = 128 -> 32, D = 256 -> 16
            if ((int64_t)((g.max_rows + rows - 1) / rows) * n_lanes < 4096) rows = 16;      // (a launch of few tiles: 4 x the workgroups)
        }
        switch (rows) {
            case 16: launch_gather_v4<16>(s, gk, g.max_rows, d_lanes, n_lanes, copy_range); break;
            case 32: launch_gather_v4<32>(s, gk, g.max_rows, d_lanes, n_lanes, copy_range); break;
            case 128: launch_gather_v4<128>(s, gk, g.max_rows, d_lanes, n_lanes, copy_range); break;
            case 256: launch_gather_v4<256>(s, gk, g.max_rows, d_lanes, n_lanes, copy_range); break;
            default: launch_gather_v4<64>(s, gk, g.max_rows, d_lanes, n_lanes, copy_range); break;
        }
    } else if (g.D > 4) {
        // rows that are not multiples of 16 bytes (D = 602: 2408-byte rows): 16-byte chunks at dword alignment + a scalar
        // tail, instead of the 8- / 4-byte vector paths of rounds 1-2 (0.65 of peak at D = 602)
        typedef float v4u __attribute__((ext_vector_type(4), aligned(4)));
        const bool small = (int64_t)((g.max_rows + LG_GATHER_ROWS - 1) / LG_GATHER_ROWS) * n_lanes < 4096;
        const bool r16 = small || (int64_t)g.D * 4 * 64 > 65536;
        const dim3 gr = r16 ? dim3(gather_grid_x(g.max_rows, 16, n_lanes), n_lanes) : grid;
        if (r16 && g.last_op) gather_kernel<v4u, 16, LG_GATHER_UNROLL, true, true><<<gr, LG_GATHER_THREADS, 0, s>>>(gk, d_lanes, copy_range);
        else if (r16) gather_kernel<v4u, 16, LG_GATHER_UNROLL, true, false><<<gr, LG_GATHER_THREADS, 0, s>>>(gk, d_lanes, copy_range);
        else if (g.last_op) gather_kernel<v4u, LG_GATHER_ROWS, LG_GATHER_UNROLL, true, true><<<gr, LG_GATHER_THREADS, 0, s>>>(gk, d_lanes, copy_range);
        else gather_kernel<v4u, LG_GATHER_ROWS, LG_GATHER_UNROLL, true, false><<<gr, LG_GATHER_THREADS, 0, s>>>(gk, d_lanes, copy_range);
    } else if (g.last_op)
        gather_kernel<float><<<grid, LG_GATHER_THREADS, 0, s>>>(gk, d_lanes, copy_range);
    else
        gather_kernel<float, LG_GATHER_ROWS, LG_GATHER_UNROLL, false, false><<<grid, LG_GATHER_THREADS, 0, s>>>(gk, d_lanes, copy_range);
    hipCheckError();
}

void launch_gather(hipStream_t s, const GatherParams& g, const LanePtrs* d_lanes, int32_t n_lanes)
{
    launch_gather_impl(s, g, d_lanes, n_lanes, true);
}

// stand-alone form (tests, probes): explicit arrays; a one-lane descriptor is staged on the stream
void launch_gather_explicit(hipStream_t s, const GatherParams& g, const int32_t* sampled_ids,
                            int32_t* cache_index_out, const int32_t* range, float* dst, int32_t dst_rows)
{
    static thread_local LanePtrs* d_lane = nullptr;
    if (d_lane == nullptr) HIP_CALL(hipMalloc(&d_lane, sizeof(LanePtrs)));
    LanePtrs h;
    memset(&h, 0, sizeof(h));
    h.sampled_ids = const_cast<int32_t*>(sampled_ids);
    h.cache_search_buffer = cache_index_out;
    h.node_counter = const_cast<int32_t*>(range);
    h.float_features = dst;
    h.feature_rows = dst_rows;
    GatherParams ge = g;
    ge.hop = -1;
    ge.first_hop = -1;
    ge.last_op = true;
    HIP_CALL(hipMemcpyAsync(d_lane, &h, sizeof(h), hipMemcpyHostToDevice, s));   // pageable source: staged before return
    launch_gather_impl(s, ge, d_lane, 1, false);
}

}  // namespace lg
