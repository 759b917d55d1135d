// kernels_sample.hip -- the multi-hop CSR neighbour sampler for gfx950 (CDNA4, wave64).
//
// What it computes is the reference's per-hop pipeline (SS = sampling_server/src):
//   batch_generate   SS/engine/operator_impl.cu:27-55
//   counter_update   SS/engine/operator_impl.cu:57-89   (folded into the kernels below)
//   FindTopo         SS/cache/cache.cu:217-225          (folded into sample_kernel)
//   random_sample    SS/engine/operator_impl.cu:175-281 (pre_sample :301-397 with is_presc)
//   construct_graph  SS/engine/operator_impl.cu:283-296
//   ClearPosMap      SS/engine/operator_impl.cu:542-548
// How it computes it is new.  The reference compacts edges/new nodes with shared + global
// atomicAdd (order is a race) and reads its counters back to the host twice per hop.  Here:
//   * sample_kernel: one lane per output slot, a 256-slot tile per workgroup; the tile's
//     frontier rows (src id, CSR row start/degree, column base) are staged once in LDS; the
//     minstd draw is a table-driven modular power + one IEEE double divide; the neighbour is
//     published with atomicMin(position_state[dst], PENDING + slot) so that the LOWEST slot
//     owns a first touch (deterministic, unlike atomicOr on a bitmap);
//   * flag_count_kernel: wave ballots count valid edges / first touches per tile;
//   * scan_kernel (one workgroup): exclusive prefix over tiles + the whole counter_update state
//     machine, so no host round trip and no <<<1,1>>> launches;
//   * scatter_kernel: ballot + mbcnt prefix inside the tile -> slot-ordered compaction of edges
//     (global ids + the frontier's local position) and of new nodes;
//   * localise_kernel: agg_src_off[e] = position of the sampled neighbour.
// Every kernel is a fixed-size grid that strides over tiles and reads the frontier length from
// device memory, so the whole hop is enqueued without knowing any size on the host.
//
// Roofline: HBM-bound irregular gather (4-byte column reads, 4-byte atomics); no MFMA.
#include "legion_core.h"

namespace lg {

// ------------------------------------------------------------------------------------------
// minstd_rand (48271^n mod 2^31-1) by three power tables: n = n0 + 2^11 n1 + 2^22 n2.
// ------------------------------------------------------------------------------------------
static constexpr uint32_t kM31 = 2147483647u;

__host__ __device__ constexpr uint32_t mulmod31(uint32_t a, uint32_t b)
{
    uint64_t p = (uint64_t)a * (uint64_t)b;           // < 2^62
    uint64_t s = (p & kM31) + (p >> 31);              // 2^31 == 1 (mod M)  -> < 2^32
    s = (s & kM31) + (s >> 31);                       // <= 2^31
    return (uint32_t)(s >= kM31 ? s - kM31 : s);
}

struct PowTables {
    uint32_t t0[2048];   // 48271^i
    uint32_t t1[2048];   // 48271^(i * 2^11)
    uint32_t t2[1024];   // 48271^(i * 2^22)
};

static constexpr PowTables make_pow_tables()
{
    PowTables t{};
    uint32_t v = 1;
    for (int i = 0; i < 2048; i++) { t.t0[i] = v; v = mulmod31(v, 48271u); }
    const uint32_t step1 = v;                          // 48271^2048
    v = 1;
    for (int i = 0; i < 2048; i++) { t.t1[i] = v; v = mulmod31(v, step1); }
    const uint32_t step2 = v;                          // 48271^(2^22)
    v = 1;
    for (int i = 0; i < 1024; i++) { t.t2[i] = v; v = mulmod31(v, step2); }
    return t;
}

__device__ const PowTables g_pow = make_pow_tables();

__device__ __forceinline__ uint32_t minstd_pow(uint32_t n)
{
    uint32_t x = mulmod31(g_pow.t0[n & 2047u], g_pow.t1[(n >> 11) & 2047u]);
    return mulmod31(x, g_pow.t2[n >> 22]);
}

// thrust::uniform_int_distribution<int>(0, deg-1) over minstd_rand, see oracle/legion_oracle.c.
__device__ __forceinline__ int32_t draw_from_x(uint32_t x, int32_t deg)
{
    double r = (double)(uint32_t)(x - 1u);
    r /= 2147483646.0;                                 // IEEE divide (no fast-math in this build)
    return (int32_t)(r * (((double)(deg - 1) + 1.0) - 0.0) + 0.0);
}

__global__ void draw_batch_kernel(const int32_t* idx, const int32_t* deg, int32_t* out, int32_t n)
{
    int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = draw_from_x(minstd_pow((uint32_t)idx[i] + 1u), deg[i]);
}

void launch_draw_batch(hipStream_t s, const int32_t* idx, const int32_t* deg, int32_t* out, int32_t n)
{
    if (n <= 0) return;
    draw_batch_kernel<<<(n + 255) / 256, 256, 0, s>>>(idx, deg, out, n);
    hipCheckError();
}

// ------------------------------------------------------------------------------------------
// batch_generate + counter_update(0)
// ------------------------------------------------------------------------------------------
__global__ void batch_generate_kernel(int32_t* __restrict__ batch_ids, int32_t* __restrict__ labels,
                                      int32_t size, int32_t counter,
                                      const int32_t* __restrict__ all_ids,
                                      const int32_t* __restrict__ all_labels, int32_t total_cap,
                                      int32_t* __restrict__ position_map, int32_t* __restrict__ nc,
                                      int32_t* __restrict__ ec, int32_t hop_num,
                                      const int32_t* __restrict__ iter_state)
{
    const int32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (iter_state != nullptr) counter = iter_state[0];   // graph replay: iteration lives on the device
    if (idx < 16) {                    // memset of both counter blocks, operator_impl.cu:155-156,
        int32_t v = 0;                 // then counter_update(op 0), :64-68
        if (idx == 1) v = size;
        if (idx == INTRABATCH_CON * 3) v = size;
        if (idx == INTRABATCH_CON * 3 - 1) v = hop_num;
        nc[idx] = v;
        ec[idx] = 0;
    }
    if (idx < size) {
        const int64_t at = (int64_t)size * counter + idx;     // kernel receives `size` as batch_size (:162)
        if (at >= total_cap) {
            batch_ids[idx] = -1;
            labels[idx] = -1;
        } else {
            const int32_t src_id = all_ids[at % total_cap];
            batch_ids[idx] = src_id;
            atomicMin(position_map + src_id, idx);            // seeds are unique (":26 assume no duplicate")
            labels[idx] = all_labels[at % total_cap];
        }
    }
}

void launch_batch_generate(hipStream_t s, int32_t* batch_ids, int32_t* labels, int32_t size,
                           int32_t counter, const int32_t* all_ids, const int32_t* all_labels,
                           int32_t total_cap, int32_t* position_map, int32_t* node_counter,
                           int32_t* edge_counter, int32_t hop_num, const int32_t* iter_state)
{
    const int32_t n = size > 16 ? size : 16;
    batch_generate_kernel<<<(n + 255) / 256, 256, 0, s>>>(batch_ids, labels, size, counter, all_ids,
                                                         all_labels, total_cap, position_map,
                                                         node_counter, edge_counter, hop_num, iter_state);
    hipCheckError();
}

// ------------------------------------------------------------------------------------------
// hop geometry shared by the three pre-scan kernels: read from the live counters
// ------------------------------------------------------------------------------------------
struct HopGeom {
    const int32_t* frontier;
    int32_t frontier_len;
    int32_t total;     // slots
    int32_t ntiles;
};

__device__ __forceinline__ HopGeom hop_geometry(const SampleArgs& a)
{
    HopGeom g;
    if (a.op_id == INTRABATCH_CON) {            // operator_impl.cu:201-203
        g.frontier = a.sampled_ids;
        g.frontier_len = a.node_counter[1];
    } else {                                    // :204-207
        g.frontier = a.agg_src_ids + a.edge_counter[0];
        g.frontier_len = a.edge_counter[1];
    }
    int64_t total = (int64_t)(g.frontier_len > 0 ? g.frontier_len : 0) * a.count;
    if (total > a.max_slots) total = a.max_slots;   // never true for a pool sized by server.cu:187-199
    g.total = (int32_t)total;
    g.ntiles = (g.total + LG_TILE - 1) / LG_TILE;
    return g;
}

// ------------------------------------------------------------------------------------------
// K1: sample.  LDS holds the tile's frontier rows.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(LG_TILE) void sample_kernel(SampleArgs a)
{
    __shared__ int64_t s_start[LG_TILE];
    __shared__ const int32_t* s_col[LG_TILE];
    __shared__ int32_t s_deg[LG_TILE];
    __shared__ int32_t s_src[LG_TILE];

    const HopGeom g = hop_geometry(a);
    const int32_t tid = threadIdx.x;
    const int32_t count = a.count;
    const bool use_topo_cache = (!a.is_presc) && (a.edge_index_map != nullptr);

    for (int32_t tile = blockIdx.x; tile < g.ntiles; tile += gridDim.x) {
        const int32_t idx0 = tile * LG_TILE;
        const int32_t last = min(idx0 + LG_TILE - 1, g.total - 1);
        const int32_t j0 = idx0 / count;
        const int32_t nsrc = last / count - j0 + 1;          // <= LG_TILE

        // stage the frontier rows of this tile (FindTopo + row header), one lane per row
        for (int32_t t = tid; t < nsrc; t += LG_TILE) {
            const int32_t j = j0 + t;
            const int32_t src = g.frontier[j];
            int32_t owner = CACHEMISS_FLAG, off = CACHEMISS_FLAG, deg = 0;
            int64_t start = 0;
            const int32_t* col = nullptr;
            if (src >= 0) {
                if (use_topo_cache) {
                    owner = (int32_t)a.edge_index_map[src];
                    off = a.edge_offset_map[src];
                }
                const int32_t slot = owner < 0 ? a.partition_count : owner;   // :224-230
                const int32_t row = owner < 0 ? src : off;
                const int64_t* ip = a.csr_node_index[slot];
                start = ip[row];
                deg = (int32_t)(ip[row + 1] - start);
                col = a.csr_dst_node_ids[slot];
            }
            if (!a.is_presc) {                 // the FindTopo outputs (hit mask = part_ind >= 0)
                a.tmp_part_ind[j] = (char)owner;
                a.tmp_part_off[j] = off;
            }
            s_start[t] = start;
            s_col[t] = col;
            s_deg[t] = deg;
            s_src[t] = src;
        }
        __syncthreads();

        const int32_t idx = idx0 + tid;
        if (idx < g.total) {
            const int32_t q = idx / count;
            const int32_t k = idx - q * count;
            const int32_t t = q - j0;
            const int32_t deg = s_deg[t];
            int32_t out = -1;
            if (k < deg) {                                            // :232-233 (src < 0 has deg 0)
                const uint32_t x = minstd_pow((uint32_t)idx + 1u);    // discard(idx) + one draw
                const int32_t pick = draw_from_x(x, deg);             // :235-238
                const int32_t dst = s_col[t][s_start[t] + (int64_t)pick];   // :239-243
                if (dst >= 0) {                                       // :244
                    out = dst;
                    atomicMin(a.position_map + dst, LG_POS_PENDING + idx);
                    if (a.edge_access_time) atomicAdd(a.edge_access_time + s_src[t], 1ull);   // :358
                }
            }
            a.slot_dst[idx] = out;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// K2: per-tile counts of valid edges and first touches
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(LG_TILE) void flag_count_kernel(SampleArgs a)
{
    __shared__ int32_t s_cnt[2][LG_TILE / 64];
    const HopGeom g = hop_geometry(a);
    const int32_t tid = threadIdx.x;
    const int32_t wave = tid >> 6, lane = tid & 63;

    for (int32_t tile = blockIdx.x; tile < g.ntiles; tile += gridDim.x) {
        const int32_t idx = tile * LG_TILE + tid;
        bool valid = false, first = false;
        if (idx < g.total) {
            const int32_t v = a.slot_dst[idx];
            valid = v >= 0;
            if (valid) {
                first = a.position_map[v] == LG_POS_PENDING + idx;
                if (first) a.slot_dst[idx] = v | (int32_t)0x80000000;
            }
        }
        const unsigned long long mv = __ballot(valid);
        const unsigned long long mf = __ballot(first);
        if (lane == 0) {
            s_cnt[0][wave] = __popcll(mv);
            s_cnt[1][wave] = __popcll(mf);
        }
        __syncthreads();
        if (tid < 2) {
            int32_t c = 0;
            for (int w = 0; w < LG_TILE / 64; w++) c += s_cnt[tid][w];
            a.tile_counts[2 * tile + tid] = c;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// K3: one workgroup: exclusive prefix over tiles, hop scratch, counter_update(op) and the
//     copy the following gather op would make (counter_update(op+1)).
// ------------------------------------------------------------------------------------------
#define LG_SCAN_THREADS 1024
__global__ __launch_bounds__(LG_SCAN_THREADS) void scan_kernel(SampleArgs a)
{
    __shared__ int32_t s_e[LG_SCAN_THREADS];
    __shared__ int32_t s_n[LG_SCAN_THREADS];
    const HopGeom g = hop_geometry(a);
    int32_t* nc = a.node_counter;
    int32_t* ec = a.edge_counter;
    const int32_t nc0 = nc[0], nc1 = nc[1], ec0 = ec[0], ec1 = ec[1];
    const int32_t tid = threadIdx.x;

    const int32_t per = (g.ntiles + LG_SCAN_THREADS - 1) / LG_SCAN_THREADS;
    const int32_t lo = min(tid * per, g.ntiles), hi = min(lo + per, g.ntiles);
    int32_t se = 0, sn = 0;
    for (int32_t t = lo; t < hi; t++) { se += a.tile_counts[2 * t]; sn += a.tile_counts[2 * t + 1]; }
    s_e[tid] = se;
    s_n[tid] = sn;
    __syncthreads();
    for (int32_t d = 1; d < LG_SCAN_THREADS; d <<= 1) {        // inclusive Hillis-Steele
        int32_t ve = 0, vn = 0;
        if (tid >= d) { ve = s_e[tid - d]; vn = s_n[tid - d]; }
        __syncthreads();
        s_e[tid] += ve;
        s_n[tid] += vn;
        __syncthreads();
    }
    int32_t pe = s_e[tid] - se, pn = s_n[tid] - sn;            // exclusive prefix of this chunk
    for (int32_t t = lo; t < hi; t++) {
        a.tile_prefix[2 * t] = pe;
        a.tile_prefix[2 * t + 1] = pn;
        pe += a.tile_counts[2 * t];
        pn += a.tile_counts[2 * t + 1];
    }
    if (tid == 0) {
        const int32_t n_edge = s_e[LG_SCAN_THREADS - 1];
        const int32_t n_new = s_n[LG_SCAN_THREADS - 1];
        int32_t* hs = a.hop_scratch;
        hs[HS_FRONTIER_IS_SEEDS] = (a.op_id == INTRABATCH_CON) ? 1 : 0;
        hs[HS_FRONTIER_OFF] = (a.op_id == INTRABATCH_CON) ? 0 : ec0;
        hs[HS_FRONTIER_LEN] = g.frontier_len;
        hs[HS_NODE_BASE] = nc0 + nc1;                          // operator_impl.cu:268
        hs[HS_EDGE_BASE] = ec0 + ec1;                          // :275
        hs[HS_N_NEW] = n_new;
        hs[HS_N_EDGE] = n_edge;
        hs[HS_SLOTS] = g.total;
        // counter_update(op_id), op_id % 3 == 0: operator_impl.cu:69-82, with nc[6] = n_new and
        // ec[2] = n_edge being what the reference's atomicAdds (:263-264) leave there
        const int32_t h = a.op_id / INTRABATCH_CON;
        nc[0] = nc0 + nc1;
        nc[1] = n_new;
        nc[INTRABATCH_CON * 2] = 0;
        nc[INTRABATCH_CON * 2 + 1] = nc0 + nc1 + n_new;
        ec[0] = ec0 + ec1;
        ec[1] = n_edge;
        ec[2] = 0;
        nc[INTRABATCH_CON * 3 + h] = nc0 + nc1 + n_new;
        ec[INTRABATCH_CON * 3 + h] = ec0 + ec1 + n_edge;
    }
}

// ------------------------------------------------------------------------------------------
// K4: slot-ordered compaction
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(LG_TILE) void scatter_kernel(SampleArgs a)
{
    __shared__ int32_t s_cnt[2][LG_TILE / 64];
    const int32_t* hs = a.hop_scratch;
    const int32_t total = hs[HS_SLOTS];
    const int32_t ntiles = (total + LG_TILE - 1) / LG_TILE;
    const bool seeds = hs[HS_FRONTIER_IS_SEEDS] != 0;
    const int32_t f_off = hs[HS_FRONTIER_OFF];
    const int32_t node_base = hs[HS_NODE_BASE], edge_base = hs[HS_EDGE_BASE];
    const int32_t* frontier = seeds ? a.sampled_ids : a.agg_src_ids + f_off;
    const int32_t tid = threadIdx.x;
    const int32_t wave = tid >> 6, lane = tid & 63;
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));

    for (int32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int32_t idx = tile * LG_TILE + tid;
        int32_t v = -1;
        if (idx < total) v = a.slot_dst[idx];
        const bool valid = v != -1;
        const bool first = valid && v < 0;
        const int32_t dst = v & 0x7FFFFFFF;
        const unsigned long long mv = __ballot(valid);
        const unsigned long long mf = __ballot(first);
        if (lane == 0) {
            s_cnt[0][wave] = __popcll(mv);
            s_cnt[1][wave] = __popcll(mf);
        }
        __syncthreads();
        int32_t we = 0, wn = 0;
        for (int w = 0; w < wave; w++) { we += s_cnt[0][w]; wn += s_cnt[1][w]; }
        if (valid) {
            const int32_t e = edge_base + a.tile_prefix[2 * tile] + we + __popcll(mv & lt);
            const int32_t q = idx / a.count;
            a.agg_src_ids[e] = dst;                            // :256, :276
            a.agg_dst_ids[e] = frontier[q];                    // :257, :277
            // position of the node sampled for == construct_graph's position_map[agg_dst_ids[e]]
            a.agg_dst_off[e] = seeds ? q : a.agg_src_off[f_off + q];
            if (first) {
                const int32_t n = node_base + a.tile_prefix[2 * tile + 1] + wn + __popcll(mf & lt);
                a.sampled_ids[n] = dst;                        // :270
                a.position_map[dst] = n;                       // :271
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// K5: construct_graph's neighbour side
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void localise_kernel(SampleArgs a)
{
    const int32_t* hs = a.hop_scratch;
    const int32_t n_edge = hs[HS_N_EDGE], edge_base = hs[HS_EDGE_BASE];
    for (int32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < n_edge; e += gridDim.x * blockDim.x)
        a.agg_src_off[edge_base + e] = a.position_map[a.agg_src_ids[edge_base + e]];   // :289-293
}

void launch_random_sample(hipStream_t s, const SampleArgs& a)
{
    // fixed grids that stride over tiles: enough workgroups to fill 256 CUs x 8, never more
    // than the hop can use
    int32_t max_tiles = (a.max_slots + LG_TILE - 1) / LG_TILE;
    if (max_tiles < 1) max_tiles = 1;
    const int32_t grid = max_tiles < 2048 ? max_tiles : 2048;
    sample_kernel<<<grid, LG_TILE, 0, s>>>(a);
    hipCheckError();
    flag_count_kernel<<<grid, LG_TILE, 0, s>>>(a);
    hipCheckError();
    scan_kernel<<<1, LG_SCAN_THREADS, 0, s>>>(a);
    hipCheckError();
    scatter_kernel<<<grid, LG_TILE, 0, s>>>(a);
    hipCheckError();
    int32_t lgrid = (a.max_slots + 255) / 256;
    if (lgrid < 1) lgrid = 1;
    if (lgrid > 2048) lgrid = 2048;
    localise_kernel<<<lgrid, 256, 0, s>>>(a);
    hipCheckError();
}

// ------------------------------------------------------------------------------------------
// end of batch: restore the untouched state for every node of the batch (the reference zeroes
// position_map in train mode only, operator_impl.cu:542-548; here the state array doubles as the
// accessed bitmap, so it is restored in every mode instead of memsetting N/8 bytes per batch).
// ------------------------------------------------------------------------------------------
__global__ void clear_pos_map_kernel(int32_t* __restrict__ position_map,
                                     const int32_t* __restrict__ sampled_ids,
                                     const int32_t* __restrict__ nc, int32_t* __restrict__ iter_state)
{
    // last kernel of a batch: advance the device-resident iteration for the next graph replay
    if (iter_state != nullptr && blockIdx.x == 0 && threadIdx.x == 0) iter_state[0] += iter_state[1];
    const int32_t hop_num = nc[INTRABATCH_CON * 3 - 1];
    const int32_t total = nc[INTRABATCH_CON * 3 + hop_num];
    for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int32_t id = sampled_ids[i];
        if (id >= 0) position_map[id] = LG_POS_UNTOUCHED;
    }
}

void launch_clear_pos_map(hipStream_t s, int32_t* position_map, const int32_t* sampled_ids,
                          const int32_t* node_counter, int32_t* iter_state)
{
    clear_pos_map_kernel<<<1024, 256, 0, s>>>(position_map, sampled_ids, node_counter, iter_state);
    hipCheckError();
}

// SS/cache/cache_impl.cuh:190-198
__global__ void hotness_measure_kernel(const int32_t* __restrict__ ids, const int32_t* __restrict__ nc,
                                       unsigned long long* __restrict__ access_map)
{
    const int32_t n = nc[INTRABATCH_CON * 2 + 1];
    for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int32_t cid = ids[i];
        if (cid >= 0) atomicAdd(access_map + cid, 1ull);
    }
}

void launch_hotness_measure(hipStream_t s, const int32_t* sampled_ids, const int32_t* node_counter,
                            unsigned long long* access_map)
{
    hotness_measure_kernel<<<1024, 256, 0, s>>>(sampled_ids, node_counter, access_map);
    hipCheckError();
}

// the bcht::find contract on the direct-mapped tables (SS/include/hashmap/bcht.hpp:105-165)
__global__ void find_kernel(const int32_t* __restrict__ keys, int32_t n,
                            const int32_t* __restrict__ map32, const char* __restrict__ map8,
                            int32_t* __restrict__ out32, char* __restrict__ out8)
{
    for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int32_t k = keys[i];
        if (out32) out32[i] = (k >= 0 && map32) ? map32[k] : CACHEMISS_FLAG;
        if (out8) out8[i] = (k >= 0 && map8) ? map8[k] : (char)CACHEMISS_FLAG;
    }
}

void launch_find(hipStream_t s, const int32_t* keys, int32_t n, const int32_t* map32, const char* map8,
                 int32_t* out32, char* out8)
{
    if (n <= 0) return;
    int32_t grid = (n + 255) / 256;
    if (grid > 2048) grid = 2048;
    find_kernel<<<grid, 256, 0, s>>>(keys, n, map32, map8, out32, out8);
    hipCheckError();
}

}  // namespace lg
