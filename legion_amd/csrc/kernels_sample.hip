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
// atomicAdd (order is a race), marks first touches in an N-bit map + an N-entry position map that it clears per batch, and
// reads its counters back to the host twice per hop.  Here a hop is
//   * sample_kernel: a workgroup owns 1024 consecutive slots, four per lane; the tile's frontier row
//     headers (row start, degree, CSR slot) are staged once in LDS; the minstd draw is a table-driven
//     modular power + one IEEE double divide; the picks (claims: vertex, slot) are written grouped by hash bucket into one
//     list per bucket of the lane (8 / 16 buckets: from the registers; 64: staged per super tile in LDS; 256: by place_kernel);
//   * dedup_lists_kernel: a (lane, bucket)'s claims de-duplicated in an LDS table against the batch's known vertices: the LOWEST
//     slot owns a first touch (deterministic, unlike atomicOr on a bitmap); every other claim's slot is marked a loser.  No
//     per-vertex state in memory at all: nothing to clear, nothing that scales with the graph.  5, 10 or 20 claims per thread in
//     registers, by how many claims PreSC saw per bucket in the last hop (operators.hip);
//   * compact_kernel: ONE pass -- tiles handed out by ticket, counts of valid edges / first touches by wave ballots, the
//     prefix over a lane's tiles by decoupled look-back, then the slot-ordered compaction of edges (global ids + both local
//     positions) and of new nodes, + the next hop's row headers; its last workgroup does the counter_update state machine,
//     so there is no host round trip and no <<<1,1>>> launch;
//   * list_known_kernel (every hop but the last): the hop's new nodes appended to the buckets' known lists.
// (Rounds 1-4 also carried two atomics forms of the first-touch state -- a uint32[N] array per lane and an open-addressing table,
// claimed with atomicMin per pick -- which the LDS form beat by 5-14 % wherever it applied; since round 5 it applies to hops of
// any size and the other two are gone: DESIGN_HISTORY.md 3, 4.2.)
// Every kernel runs with grid.y = lanes (independent mini-batches, LanePtrs) and takes its pointers
// from the lane descriptor as global-address-space pointers.
// Every kernel is a fixed-size grid that strides over tiles and reads the frontier length from
// device memory, so the whole hop is enqueued without knowing any size on the host.
//
// Bound: sample_kernel by the part's rate of random 128-byte requests; the other kernels by their dependent chains
// (DESIGN.md 4.2); no MFMA.
#include "legion_core.h"

#include <cstdlib>

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

// the per-batch buffers the two bracket kernels touch, as global-address-space pointers
struct BracketLane {
    LG_G int32_t* sampled_ids; LG_G int32_t* labels; LG_G int32_t* node_counter; LG_G int32_t* edge_counter;
    LG_G int32_t* hop_scratch; LG_G int32_t* node_slot;
    LG_G int32_t* known_cnt; LG_G int32_t* claim_cnt; int32_t lds_buckets;
    LG_G int32_t* counter_mirror;
};
__device__ __forceinline__ BracketLane bracket_lane(const LanePtrs& P)
{
    BracketLane L;
    L.sampled_ids = LG_GPTR(int32_t, P.sampled_ids); L.labels = LG_GPTR(int32_t, P.labels);
    L.node_counter = LG_GPTR(int32_t, P.node_counter); L.edge_counter = LG_GPTR(int32_t, P.edge_counter);
    L.hop_scratch = LG_GPTR(int32_t, P.hop_scratch); L.node_slot = LG_GPTR(int32_t, P.node_slot);
    L.known_cnt = LG_GPTR(int32_t, P.known_cnt); L.claim_cnt = LG_GPTR(int32_t, P.claim_cnt); L.lds_buckets = P.lds_buckets;
    L.counter_mirror = LG_GPTR(int32_t, P.counter_mirror);
    return L;
}

__device__ __forceinline__ void raise_error(LG_G int32_t* hop_scratch, LG_G int32_t* err_flag, int32_t bits)
{
    __hip_atomic_fetch_or(hop_scratch + HS_ERROR, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (err_flag) __hip_atomic_fetch_or(err_flag, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ------------------------------------------------------------------------------------------
// batch_generate + counter_update(0)
// ------------------------------------------------------------------------------------------
__global__ void batch_generate_kernel(SeedParams p, const LanePtrs* __restrict__ lanes)
{
    const BracketLane L = bracket_lane(lanes[blockIdx.y]);
    const int32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    // lane i of a group takes iteration base + i; under graph replay the base lives on the device
    const int32_t counter = (p.iter_state != nullptr ? p.iter_state[0] : p.counter0) + (int32_t)blockIdx.y;
    // operator_impl.cu:159 -- the clamped last batch (may be <= 0: nothing is sampled)
    const int32_t size = ((int64_t)p.batch_size * (counter + 1) >= p.total_cap)
                             ? (p.total_cap - p.batch_size * counter) : p.batch_size;
    if (L.known_cnt != nullptr && idx < L.lds_buckets) L.known_cnt[idx] = 0;    // lds form: the batch's known lists start empty
    if (L.claim_cnt != nullptr && idx < L.lds_buckets) L.claim_cnt[idx * LG_CLAIM_CNT_STRIDE] = 0;    // (every hop's de-duplication leaves them zero; a batch cut short by an error may not)
    if (idx < 16) {                    // memset of both counter blocks, operator_impl.cu:155-156,
        int32_t v = 0;                 // then counter_update(op 0), :64-68
        if (idx == 1) v = size;
        if (idx == INTRABATCH_CON * 3) v = size;
        if (idx == INTRABATCH_CON * 3 - 1) v = p.hop_num;
        L.node_counter[idx] = v;
        L.edge_counter[idx] = 0;
        if (idx == 0) {                // range snapshot for the seeds' gather (op 1)
            L.hop_scratch[HS_RANGE] = 0;
            L.hop_scratch[HS_RANGE + 1] = size;
        }
    }
    if (idx < size) {
        const int64_t at = (int64_t)size * counter + idx;     // kernel receives `size` as batch_size (:162)
        if (at >= p.total_cap) {
            L.sampled_ids[idx] = -1;
            L.labels[idx] = -1;
        } else {
            const int32_t src_id = p.all_ids[at % p.total_cap];
            L.sampled_ids[idx] = src_id;
            if (L.node_slot != nullptr) L.node_slot[idx] = LG_FS_UNKNOWN;     // seeds: the gather looks their cache slots up
            // (no per-vertex state: the de-duplication re-reads the seeds from sampled_ids; ":26 assume no duplicate")
            L.labels[idx] = p.all_labels[at % p.total_cap];
        }
    }
}

void launch_batch_generate(hipStream_t s, const SeedParams& p, const LanePtrs* d_lanes, int32_t n_lanes)
{
    const int32_t n = p.batch_size > 256 ? p.batch_size : 256;      // (>= the lane's hash buckets: their list counts are zeroed here)
    batch_generate_kernel<<<dim3((n + 255) / 256, n_lanes), 256, 0, s>>>(p, d_lanes);
    hipCheckError();
}

// ------------------------------------------------------------------------------------------
// hop geometry shared by the three pre-scan kernels: read from the live counters
// ------------------------------------------------------------------------------------------
// the kernels' view of one lane: the launch-wide hop parameters + that lane's buffers
typedef int32_t lg_v4i __attribute__((ext_vector_type(4)));
typedef int32_t lg_v2i __attribute__((ext_vector_type(2)));
template <int NB> __device__ __forceinline__ int64_t lg_claim_at(int32_t b, int32_t k)      // entry k of bucket b's claim list (LanePtrs)
{
    return ((((int64_t)(k >> LG_CLAIM_CHUNK_BITS) * NB) + b) << LG_CLAIM_CHUNK_BITS) + (k & (LG_CLAIM_CHUNK - 1));
}
struct SampleArgs {
    int32_t op_id, count, partition_count, max_slots;
    int32_t* const* csr_dst_node_ids;
    int32_t* const* csr_dst_x;
    const LG_G int32_t* col_full; const LG_G lg_v2i* colx_full;
    const LG_G RowHdr* row_hdr;
    bool last_hop, is_presc;
    LG_G unsigned long long* edge_access_time;
    LG_G unsigned long long* topo_transactions;
    // the lane's buffers, in the global address space (see LG_G in legion_core.h)
    LG_G int32_t* sampled_ids; LG_G int32_t* agg_src_ids; LG_G int32_t* agg_dst_ids; LG_G int32_t* agg_src_off; LG_G int32_t* agg_dst_off;
    LG_G char* tmp_part_ind; LG_G int32_t* node_counter; LG_G int32_t* edge_counter;
    LG_G int32_t* slot_dst; LG_G int32_t* slot_pos; LG_G int32_t* slot_fs; LG_G int32_t* node_slot; LG_G unsigned long long* tile_state; LG_G int32_t* hop_scratch;
    LG_G RowHdr* fh_edge;
    LG_G int32_t* err_flag;
    LG_G unsigned long long* claim_pairs; LG_G int32_t* run_off; LG_G int32_t* claim_cnt; int32_t claim_cap, ids_cap;
    LG_G unsigned long long* known_pairs; LG_G int32_t* known_cnt; int32_t known_cap;
};

// 16-byte header load / store through a global-address-space pointer (no implicit struct copy across
// address spaces in HIP C++)

__device__ __forceinline__ RowHdr load_hdr(const LG_G RowHdr* p)
{
    const lg_v4i t = *(const LG_G lg_v4i*)p;
    RowHdr h;
    h.start = (int64_t)(((uint64_t)(uint32_t)t.y << 32) | (uint32_t)t.x);
    h.deg = t.z;
    h.slot = t.w;
    return h;
}
__device__ __forceinline__ void store_hdr(LG_G RowHdr* p, const RowHdr& h)
{
    lg_v4i t;
    t.x = (int32_t)(uint32_t)h.start;
    t.y = (int32_t)((uint64_t)h.start >> 32);
    t.z = h.deg;
    t.w = h.slot;
    *(LG_G lg_v4i*)p = t;
}

__device__ __forceinline__ SampleArgs lane_args(const HopParams& p, const LanePtrs* __restrict__ lanes)
{
    const LanePtrs& L = lanes[blockIdx.y];
    SampleArgs a;
    a.op_id = p.op_id; a.count = p.count; a.partition_count = p.partition_count; a.max_slots = p.max_slots;
    a.csr_dst_node_ids = p.csr_dst_node_ids; a.csr_dst_x = p.csr_dst_x;
    a.col_full = LG_GPTR(const int32_t, p.col_full); a.colx_full = LG_GPTR(const lg_v2i, p.colx_full); a.row_hdr = LG_GPTR(const RowHdr, p.row_hdr); a.last_hop = p.last_hop; a.is_presc = p.is_presc;
    a.edge_access_time = LG_GPTR(unsigned long long, p.edge_access_time);
    a.topo_transactions = LG_GPTR(unsigned long long, p.topo_transactions);
    a.sampled_ids = LG_GPTR(int32_t, L.sampled_ids); a.agg_src_ids = LG_GPTR(int32_t, L.agg_src_ids);
    a.agg_dst_ids = LG_GPTR(int32_t, L.agg_dst_ids); a.agg_src_off = LG_GPTR(int32_t, L.agg_src_off);
    a.agg_dst_off = LG_GPTR(int32_t, L.agg_dst_off); a.tmp_part_ind = LG_GPTR(char, L.tmp_part_ind);
    a.node_counter = LG_GPTR(int32_t, L.node_counter);
    a.edge_counter = LG_GPTR(int32_t, L.edge_counter); a.slot_dst = LG_GPTR(int32_t, L.slot_dst);
    a.slot_pos = LG_GPTR(int32_t, L.slot_pos);
    a.slot_fs = LG_GPTR(int32_t, L.slot_fs); a.node_slot = LG_GPTR(int32_t, L.node_slot); a.tile_state = LG_GPTR(unsigned long long, L.tile_state); a.hop_scratch = LG_GPTR(int32_t, L.hop_scratch);
    a.fh_edge = LG_GPTR(RowHdr, L.fh_edge);
    a.err_flag = LG_GPTR(int32_t, L.err_flag);
    a.claim_pairs = LG_GPTR(unsigned long long, L.claim_pairs);
    a.run_off = LG_GPTR(int32_t, L.run_off);
    a.claim_cnt = LG_GPTR(int32_t, L.claim_cnt); a.claim_cap = L.claim_cap; a.ids_cap = L.ids_cap;
    a.known_pairs = LG_GPTR(unsigned long long, L.known_pairs); a.known_cnt = LG_GPTR(int32_t, L.known_cnt); a.known_cap = L.known_cap;
    return a;
}

struct HopGeom {
    const LG_G int32_t* frontier;
    int32_t frontier_len;
    int32_t frontier_off;   // offset of the frontier inside the per-edge arrays (0 for the seeds)
    int32_t total;          // slots
    int32_t ntiles;         // 256-slot tiles (compaction granularity)
    int32_t nsuper;         // LG_SUPER-slot super tiles (work granularity of one workgroup)
};

__device__ __forceinline__ HopGeom hop_geometry(const SampleArgs& a)
{
    HopGeom g;
    if (a.op_id == INTRABATCH_CON) {            // operator_impl.cu:201-203
        g.frontier = a.sampled_ids;
        g.frontier_len = a.node_counter[1];
        g.frontier_off = 0;
    } else {                                    // :204-207
        g.frontier_off = a.edge_counter[0];
        g.frontier = a.agg_src_ids + g.frontier_off;
        g.frontier_len = a.edge_counter[1];
    }
    int64_t total = (int64_t)(g.frontier_len > 0 ? g.frontier_len : 0) * a.count;
    if (total > a.max_slots) total = a.max_slots;   // never true for a pool sized by server.cu:187-199
    g.total = (int32_t)total;
    g.ntiles = (g.total + LG_TILE - 1) / LG_TILE;
    g.nsuper = (g.total + LG_SUPER - 1) / LG_SUPER;
    return g;
}

// ------------------------------------------------------------------------------------------
// K1: sample.  A workgroup owns a super tile of 1024 consecutive slots, four per lane (lane l of
// wave w handles slots idx0 + u*256 + 64w + l, u = 0..3), so every wave-instruction still
// covers 64 consecutive slots and each lane keeps four independent column loads and atomics in
// flight.  The super tile's frontier row headers ({start, degree, CSR slot}, 16 B) are staged in
// LDS: for hop >= 2 they were written next to the edges by the previous hop's scatter (one
// coalesced 16-byte load per frontier entry), for hop 1 they are looked up in the per-vertex
// header table here.
// ------------------------------------------------------------------------------------------
#ifndef LG_SAMPLE_SGPRS
#define LG_SAMPLE_SGPRS 80           // 8 workgroups of 256 threads per CU need <= 80 SGPRs (MI355X_MICROARCH.md, residency); 90-106 give 6-7
#endif
// !SINGLE (256-bucket class): the kernel samples K super tiles (a partition tile), counts their claims per bucket and
// reserves, with one atomic per bucket, that many places of each bucket's claim list (run_off: {first place, count} per
// partition tile and bucket); place_kernel writes the pairs (see there).  SINGLE + STAGED (64-bucket class): below.
template <int BB, bool SINGLE, bool STAGED = false>      // 2^BB hash buckets per lane; SINGLE: partition tile = super tile; STAGED: see below
__global__ __launch_bounds__(LG_TILE) __attribute__((amdgpu_num_sgpr(LG_SAMPLE_SGPRS))) void sample_kernel(HopParams hp, const LanePtrs* __restrict__ lanes)
{
    constexpr int NB = 1 << BB;
    const int32_t K = SINGLE ? 1 : hp.lds_k;      // super tiles per partition tile
    static_assert(NB <= LG_TILE, "one thread per bucket in the prefix");
    const SampleArgs a = lane_args(hp, lanes);
    __shared__ RowHdr s_hdr[LG_SUPER];
    __shared__ int32_t s_bcnt[NB], s_boff[SINGLE ? NB : 1], s_list[STAGED ? NB : 1];
    static_assert(!STAGED || (SINGLE && NB <= 64), "the staged form scans its buckets with one wave");

    const HopGeom g = hop_geometry(a);
    const int32_t tid = threadIdx.x;
    const int32_t count = a.count;
    const bool seeds = (a.op_id == INTRABATCH_CON);
    const LG_G RowHdr* fh = a.fh_edge + g.frontier_off;

    // a hop's claims go, grouped by hash bucket, to ONE LIST PER BUCKET of the lane.  SINGLE (8 / 16 / 64 buckets): ranks are
    // taken while the super tile is sampled and the pairs are written here.  Otherwise (256 buckets: a bucket's share of a super
    // tile is a pair or two) the kernel only samples and counts; place_kernel re-reads slot_dst and writes the pairs.
    const int32_t nparts = (g.nsuper + K - 1) / K;
    for (int32_t m = blockIdx.x; m < nparts; m += gridDim.x) {
        if (!SINGLE && tid < NB) s_bcnt[tid] = 0;      // (visible after the first barrier of the first super tile)
        for (int32_t sub = 0; sub < K; sub++) {
            const int32_t st = m * K + sub;
            if (st >= g.nsuper) break;
            const int32_t idx0 = st * LG_SUPER;
            const int32_t last = min(idx0 + LG_SUPER - 1, g.total - 1);
            const int32_t j0 = idx0 / count;
            const int32_t nsrc = last / count - j0 + 1;          // <= LG_SUPER

            // the draws do not depend on the frontier: start their table loads first
            uint32_t x[LG_SLOTS_PER_LANE];
#pragma unroll
            for (int u = 0; u < LG_SLOTS_PER_LANE; u++) x[u] = minstd_pow((uint32_t)(idx0 + u * LG_TILE + tid) + 1u);

            unsigned long long tx = 0;
            for (int32_t t = tid; t < nsrc; t += LG_TILE) {
                RowHdr h;
                bool real = true;
                if (seeds) {
                    const int32_t src = g.frontier[j0 + t];
                    if (src >= 0) {
                        h = load_hdr(a.row_hdr + src);
                    } else {
                        h.start = 0; h.deg = 0; h.slot = a.partition_count;
                        real = false;
                    }
                } else {
                    h = load_hdr(fh + j0 + t);
                }
                s_hdr[t] = h;
                // PreSC: what this row's topology reads cost in 64-byte transactions (row-pointer pair + the
                // sectors its picks can touch); a row is counted by the super tile its first slot falls in
                if (a.topo_transactions && real && (int64_t)(j0 + t) * count >= idx0)
                    tx += 1ull + (unsigned long long)min(count, (h.deg * 4 + 63) / 64);
                if (!a.is_presc)   // FindTopo's hit mask: owner device of the cached row, or -2 (cache.cu:217-225)
                    a.tmp_part_ind[j0 + t] = (char)(h.slot == a.partition_count ? CACHEMISS_FLAG : h.slot);
            }
            if (a.topo_transactions) {               // wave sum, one atomic per wave
                for (int off = 32; off > 0; off >>= 1) tx += __shfl_down(tx, off);
                if ((tid & 63) == 0 && tx != 0)
                    __hip_atomic_fetch_add(a.topo_transactions, tx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();

            if (SINGLE && tid < NB) s_bcnt[tid] = 0;     // (made visible by the barrier above the loads' use below)
            int32_t dst[LG_SLOTS_PER_LANE], fs[LG_SLOTS_PER_LANE];
#pragma unroll
            for (int u = 0; u < LG_SLOTS_PER_LANE; u++) {
                const int32_t idx = idx0 + u * LG_TILE + tid;
                dst[u] = -1;
                fs[u] = LG_FS_UNKNOWN;
                if (idx < g.total) {
                    const int32_t q = idx / count;
                    const int32_t k = idx - q * count;
                    const RowHdr h = s_hdr[q - j0];
                    if (k < h.deg) {                                           // :232-233 (src < 0 has deg 0)
                        const int32_t pick = draw_from_x(x[u], h.deg);         // :235-238
                        // column slots: the same sector read as 8 bytes brings the neighbour's feature-cache slot along.  Picks
                        // from the full CSR (slot P: all but the cached-topology rows) address it through pointers that came
                        // with the launch; only a cached row's pick loads its column array's address from the table first
                        const int64_t at = h.start + (int64_t)pick;
                        if (h.slot == a.partition_count) {
                            // (non-temporal: a pick brings a whole 128-byte line in for 4-8 bytes, and most lines are not picked from
                            // again before they are evicted -- tools/micro/random_load_policy.hip: 48.9 -> 53.5 G random loads/s, every
                            // allocation kind and cache-policy bit fetches the full line; whole job +1.2 % at B = 1024 and 8000)
                            if (a.colx_full != nullptr) {
                                const lg_v2i e = __builtin_nontemporal_load(&a.colx_full[at]);
                                dst[u] = e.x;
                                fs[u] = e.y;
                            } else {
                                dst[u] = __builtin_nontemporal_load(&a.col_full[at]);                        // :239-243
                            }
                        } else {
                            const LG_G lg_v2i* cx = a.csr_dst_x != nullptr ? LG_GPTR(const lg_v2i, a.csr_dst_x[h.slot]) : nullptr;
                            if (cx != nullptr) {
                                const lg_v2i e = cx[at];
                                dst[u] = e.x;
                                fs[u] = e.y;
                            } else {
                                dst[u] = LG_GPTR(const int32_t, a.csr_dst_node_ids[h.slot])[at];
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < LG_SLOTS_PER_LANE; u++) {
                const int32_t idx = idx0 + u * LG_TILE + tid;
                if (idx < g.total) {
                    // (no claim here: the pair goes to its hash bucket below / in place_kernel)
                    if (dst[u] >= 0 && a.edge_access_time)                     // :358
                        __hip_atomic_fetch_add(a.edge_access_time + g.frontier[idx / count], 1ull, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT);
                    if (dst[u] < 0) dst[u] = -1;                               // :244
                    // slot_dst / slot_fs are read once, by the compaction two kernels later: non-temporal stores (+0.5 % on the whole job,
                    // round 4, five alternating runs per build: they do not push the column arrays' lines out of the caches)
                    __builtin_nontemporal_store(dst[u], &a.slot_dst[idx]);
                    a.slot_pos[idx] = -1;      // "no position yet": compact_kernel publishes a first touch's position here (plain store: no difference)
                    if (a.slot_fs != nullptr && dst[u] >= 0) __builtin_nontemporal_store(fs[u], &a.slot_fs[idx]);     // (read for first-touch slots only)
                }
            }
            if (SINGLE && STAGED) {
                // 64 buckets: a super tile's ~450 claims are ~7 per bucket.  Written straight from the registers (as below) a wave's 64
                // claims would go to ~40 different lists: 8-byte stores all over the lane's pair array, +60 % memory requests in a
                // kernel that is bound by them.  So the claims are ranked and STAGED in LDS grouped by bucket -- in the row-header
                // stage, which nobody reads any more once every pick has been loaded: no LDS added, the kernel keeps its 8 workgroups
                // per CU -- and written out entry by entry: consecutive entries of a bucket go to consecutive places of its list, a
                // wave's store covers ~9 runs instead of ~40.  One reservation per non-empty bucket and super tile.  (Round 4 did this
                // with a second kernel over partition tiles of 8 super tiles, place_kernel: 93-107 us per 64-lane group of B = 8000
                // for 250 MB of traffic; the 256-bucket class, where a bucket's share of a super tile is 1-2 claims, still does.)
                int32_t rank[LG_SLOTS_PER_LANE], bkt[LG_SLOTS_PER_LANE];
                unsigned long long* s_stage = reinterpret_cast<unsigned long long*>(s_hdr);      // LG_SUPER pairs = 8 KB of the 16 KB
                __syncthreads();                                   // every thread has read its headers; s_bcnt zeroed
#pragma unroll
                for (int u = 0; u < LG_SLOTS_PER_LANE; u++) {
                    bkt[u] = -1;
                    if (dst[u] >= 0) {
                        bkt[u] = (int32_t)(lg_tab_hash(dst[u]) & (NB - 1));
                        rank[u] = atomicAdd(&s_bcnt[bkt[u]], 1);
                    }
                }
                __syncthreads();
                if (tid < 64) {                                    // wave 0: exclusive prefix of the bucket counts + the reservations
                    const int32_t c = tid < NB ? s_bcnt[tid] : 0;
                    int32_t inc = c;
                    for (int d = 1; d < 64; d <<= 1) { const int32_t o = __shfl_up(inc, d); if (tid >= d) inc += o; }
                    if (tid < NB) {
                        s_boff[tid] = inc - c;
                        s_list[tid] = c > 0 ? __hip_atomic_fetch_add(a.claim_cnt + tid * LG_CLAIM_CNT_STRIDE, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
                    }
                }
                __syncthreads();
#pragma unroll
                for (int u = 0; u < LG_SLOTS_PER_LANE; u++)
                    if (bkt[u] >= 0)
                        s_stage[s_boff[bkt[u]] + rank[u]] = ((unsigned long long)(uint32_t)dst[u] << 32) | (uint32_t)(idx0 + u * LG_TILE + tid);
                __syncthreads();
                const int32_t tot = s_boff[NB - 1] + s_bcnt[NB - 1];
                for (int32_t i = tid; i < tot; i += LG_TILE) {
                    const unsigned long long pr = s_stage[i];
                    const int32_t bk = (int32_t)(lg_tab_hash((int32_t)(pr >> 32)) & (NB - 1));
                    const int32_t at = s_list[bk] + (i - s_boff[bk]);
                    if (at < a.claim_cap) a.claim_pairs[lg_claim_at<NB>(bk, at)] = pr;
                }
            } else if (SINGLE) {
                // the super tile's claims, grouped by hash bucket, into one run of the lane's pair array: ranks by LDS atomics
                // (the order inside a bucket does not matter), ONE global reservation per super tile
                int32_t rank[LG_SLOTS_PER_LANE], bkt[LG_SLOTS_PER_LANE];
                __syncthreads();                                   // s_bcnt zeroed by every wave's view
#pragma unroll
                for (int u = 0; u < LG_SLOTS_PER_LANE; u++) {
                    bkt[u] = -1;
                    if (dst[u] >= 0) {
                        bkt[u] = (int32_t)(lg_tab_hash(dst[u]) & (NB - 1));
                        rank[u] = atomicAdd(&s_bcnt[bkt[u]], 1);
                    }
                }
                __syncthreads();
                // one list per bucket: the super tile's claims of a bucket take the next places of that bucket's list (one
                // reservation per bucket and super tile).  A claim past the list's capacity is not written: the count says so, and
                // the bucket's de-duplication workgroup then reads the hop's slots instead of the list
                if (tid < NB) {
                    const int32_t c = s_bcnt[tid];
                    s_boff[tid] = c > 0 ? __hip_atomic_fetch_add(a.claim_cnt + tid * LG_CLAIM_CNT_STRIDE, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
                }
                __syncthreads();
#pragma unroll
                for (int u = 0; u < LG_SLOTS_PER_LANE; u++)
                    if (bkt[u] >= 0) {
                        const int32_t at = s_boff[bkt[u]] + rank[u];
                        if (at < a.claim_cap)
                            a.claim_pairs[lg_claim_at<NB>(bkt[u], at)] =
                                ((unsigned long long)(uint32_t)dst[u] << 32) | (uint32_t)(idx0 + u * LG_TILE + tid);
                    }
            }

            if (!SINGLE) {
#pragma unroll
                for (int u = 0; u < LG_SLOTS_PER_LANE; u++)
                    if (dst[u] >= 0) atomicAdd(&s_bcnt[lg_tab_hash(dst[u]) & (NB - 1)], 1);
            }
            __syncthreads();
        }
        if (!SINGLE) {
            // one reservation per bucket and partition tile (a count past the list's capacity is not written by place_kernel:
            // the list's count then says so, and the bucket's de-duplication workgroup reads the hop's slots instead)
            if (tid < NB) {
                const int32_t c = s_bcnt[tid];
                lg_v2i r;
                r.x = c > 0 ? __hip_atomic_fetch_add(a.claim_cnt + tid * LG_CLAIM_CNT_STRIDE, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
                r.y = c;
                ((LG_G lg_v2i*)a.run_off)[(int64_t)m * NB + tid] = r;
            }
            __syncthreads();                       // the next partition tile zeroes s_bcnt
        }
    }
}

// ------------------------------------------------------------------------------------------
// K1a (256-bucket class, partition tiles of at most LG_PLACE_MAX_K super tiles): the pairs of a partition
// tile, written to the buckets' claim lists.  With 256 buckets a wave's 64 pairs go to ~60 different lists: written straight
// from the sampling kernel they are 8-byte stores all over the lane's pair array (that sweep cost the hop-3 launch of [15,10,5]
// at B = 8000 as much again as its scattered column loads).  Here the partition tile's pairs are ranked and staged in LDS grouped
// by bucket (8 KB per super tile), and the staged block is then written out entry by entry: consecutive entries of a bucket go
// to consecutive places of that bucket's list (the places sample_kernel reserved), so a bucket's share of the partition tile
// leaves as one run.  The sampling kernel keeps its occupancy for the scattered loads (no staging there); this kernel reads
// slot_dst as a plain stream.
// ------------------------------------------------------------------------------------------
// Workgroups of 1024 threads (one slot per thread and super tile): with the 64 KB stage of K = 8 two of them share a CU, i.e. 32
// waves per CU keep this stream's loads in flight (round 4's 256-thread workgroups left 8 waves per CU: 107 us per 64-lane
// group of B = 8000 for 250 MB).
#define LG_PLACE_MAX_K 8
#define LG_PLACE_THREADS 1024
template <int BB>
__global__ __launch_bounds__(LG_PLACE_THREADS) void place_kernel(HopParams hp, const LanePtrs* __restrict__ lanes)
{
    constexpr int NB = 1 << BB;
    static_assert(NB <= 256 && LG_SUPER == LG_PLACE_THREADS, "one thread per bucket in the prefix, one slot per thread and super tile");
    extern __shared__ unsigned long long s_stage[];           // hp.lds_k * LG_SUPER pairs
    __shared__ int32_t s_off[NB + 1], s_cnt[NB], s_list[NB], s_wtot[4];
    const SampleArgs a = lane_args(hp, lanes);
    const int32_t K = hp.lds_k;
    const HopGeom g = hop_geometry(a);
    const int32_t tid = threadIdx.x;
    const int32_t nparts = (g.nsuper + K - 1) / K;
    for (int32_t m = blockIdx.x; m < nparts; m += gridDim.x) {
        // what sample_kernel reserved: {first place in the bucket's list, count}; exclusive prefix of the counts = the staging order
        lg_v2i r; r.x = 0; r.y = 0;
        if (tid < NB) r = ((const LG_G lg_v2i*)a.run_off)[(int64_t)m * NB + tid];
        int32_t inc = r.y;
        if (tid < 256) {
            for (int d = 1; d < 64; d <<= 1) { const int32_t o = __shfl_up(inc, d); if ((tid & 63) >= d) inc += o; }
            if ((tid & 63) == 63) s_wtot[tid >> 6] = inc;
        }
        __syncthreads();
        int32_t wbase = 0, tot = 0;
        for (int w = 0; w < 4; w++) { if (w < (tid >> 6)) wbase += s_wtot[w]; tot += s_wtot[w]; }
        if (tid < NB) { s_off[tid] = wbase + inc - r.y; s_list[tid] = r.x; s_cnt[tid] = 0; }
        if (tid == 0) s_off[NB] = tot;
        __syncthreads();
        for (int32_t sub0 = 0; sub0 < K; sub0 += 4) {
            int32_t d[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int32_t idx = (m * K + sub0 + u) * LG_SUPER + tid;
                d[u] = (sub0 + u < K && idx < g.total) ? a.slot_dst[idx] : -1;
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (d[u] < 0) continue;
                const int32_t bk = (int32_t)(lg_tab_hash(d[u]) & (NB - 1));
                const int32_t rk = atomicAdd(&s_cnt[bk], 1);
                s_stage[s_off[bk] + rk] = ((unsigned long long)(uint32_t)d[u] << 32) | (uint32_t)((m * K + sub0 + u) * LG_SUPER + tid);
            }
        }
        __syncthreads();
        for (int32_t i = tid; i < tot; i += LG_PLACE_THREADS) {
            const unsigned long long pr = s_stage[i];
            const int32_t bk = (int32_t)(lg_tab_hash((int32_t)(pr >> 32)) & (NB - 1));
            const int32_t at = s_list[bk] + (i - s_off[bk]);
            if (at < a.claim_cap) a.claim_pairs[lg_claim_at<NB>(bk, at)] = pr;
        }
        __syncthreads();                           // (the next partition tile re-uses the stage and the counts)
    }
}

// ------------------------------------------------------------------------------------------
// K1b (lds form): de-duplication of a hop's claims, one workgroup per (bucket, lane), entirely in LDS.
//   table word = [ vertex : 32 | pending : 1 | value : 31 ], empty = all ones, ordered linear probing with atomicMin.
//   1. the batch's known vertices that hash into this bucket go in with their position: the seeds from sampled_ids, the
//      nodes earlier hops added from the bucket's known list (list_known_kernel) -- or all of them from sampled_ids when the
//      list outgrew its capacity;
//   2. the bucket's claims go in as pending | slot: per vertex the lowest value survives -- a known position beats any
//      slot, a lower slot beats a higher one;
//   3. every claim looks its vertex up: the claim that IS the surviving word is a first touch and stays unmarked; every
//      other claim's slot is marked a loser -- slot_dst[slot] = -2 - vertex (LG_SLOT_LOSER: any int32 vertex id fits, and the
//      compaction reads the mark in the stream it reads anyway) -- and gets, in slot_pos, the vertex's final position or
//      -2 - (the winning slot: always a LOWER slot of this hop, and itself a first touch).
// A bucket whose vertices cannot fit the table is processed in P passes over sub-buckets (further hash bits), so the
// result never depends on how the hash spreads the batch.  Nothing survives the hop: nothing to clear, no state that
// scales with the graph.
// The claims arrive as ONE LIST PER BUCKET in every class (8 / 16 / 64 buckets: written by sample_kernel; 256: by place_kernel).
// A workgroup's life used to be a chain of dependent round trips to memory (lane pointers -> live counters -> segment table ->
// claims); with one list per bucket, what a thread reads first -- its claims of the list, its seed, its entry of the known list --
// sits at addresses that do not depend on the live counters, so these loads leave TOGETHER with the loads of the counters and
// list lengths (entries past the live lengths are stale and are masked when they are used).
// Measured on the 512-lane group of the headline workload (round 4, timing-only builds): launching 4096 workgroups of 16 waves
// costs 54 us before any of them does anything, with all reads requested up front and one barrier it is 108 us, the table work
// brings it to ~190.  Fewer, longer-lived workgroups remove launch cost and hide the loads -- and lose under the weave, where the
// heavy stream's kernel gets its share of the machine by asking for slots again and again while the low-priority stream's
// workgroups take every slot a long-lived workgroup cannot ask for again (DESIGN_HISTORY 4.2: two buckets per workgroup, a
// persistent launch, one workgroup per lane for small hops -- all rejected and removed).
// ------------------------------------------------------------------------------------------
#ifndef LG_DEDUP_THREADS
#define LG_DEDUP_THREADS 1024
#endif

// (SGPR cap: two of these 16-wave workgroups share a CU only while the kernel stays within 80 SGPRs -- 82..96 admit 28 waves
// per CU, i.e. ONE workgroup, and the kernel takes 150 us instead of 94 at hop 2 of a 256-lane group; measured, round 3)
//
// A workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every global load AND STORE the wave has in
// flight (its fence covers global memory): loads requested for later use would be waited for at the next barrier, and a tile's
// stores would have to land before the next tile's loads could be addressed.
__device__ __forceinline__ void lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

template <int BB, int CL, int TB = LG_LDS_TABLE_BITS>      // CL: claims a thread keeps in registers (a bucket of at most CL * LG_DEDUP_THREADS claims is "resident");
                                                            // TB: log2 words of the LDS table (13: 64 KB, two workgroups per CU; 14: 128 KB, one -- the CL = 16 form, whose 85 VGPRs
                                                            // admit one 16-wave workgroup per CU anyway)
__global__ __launch_bounds__(LG_DEDUP_THREADS, (CL <= LG_DEDUP_CLAIMS_MID ? 8 : 4)) __attribute__((amdgpu_num_sgpr(80)))      // (8 waves per SIMD = two workgroups per CU: at most 64 VGPRs)
void dedup_lists_kernel(HopParams hp, const LanePtrs* __restrict__ lanes)
{
    constexpr int NB = 1 << BB;
    constexpr uint32_t PENDING = 0x80000000u;
    constexpr int TABLE = 1 << TB;
    auto slot_of = [](uint32_t h) { return (h * 0x9E3779B1u) >> (32 - TB); };
    __shared__ unsigned long long s_tab[TABLE];
    __shared__ int32_t s_full;
    const int32_t tid = threadIdx.x, b = (int32_t)blockIdx.x;
    const SampleArgs a = lane_args(hp, lanes);

    // what the workgroup reads first: the thread's claims u * THREADS + tid of the bucket's list, its entry of the bucket's known
    // list, the list lengths, its first seed and the node counts before this hop -- one round trip
    unsigned long long rp[CL];
#pragma unroll
    for (int u = 0; u < CL; u++) {
        const int32_t k = u * LG_DEDUP_THREADS + tid;
        rp[u] = k < a.claim_cap ? a.claim_pairs[lg_claim_at<NB>(b, k)] : ~0ull;
    }
    unsigned long long kl0 = (a.known_pairs != nullptr && tid < a.known_cap) ? a.known_pairs[(int64_t)b * a.known_cap + tid] : ~0ull;
    const int32_t n_listed = a.known_pairs != nullptr ? a.known_cnt[b] : 0;
    const int32_t total = a.claim_cnt[b * LG_CLAIM_CNT_STRIDE];      // (the count of the bucket's claims even when the list could not take them all)
    const int32_t kid0 = tid < a.ids_cap ? a.sampled_ids[tid] : -1;
    const int32_t n_known = a.node_counter[0] + a.node_counter[1];
    const int32_t n_seed = min(max(a.node_counter[INTRABATCH_CON * 3], 0), n_known);

    auto insert = [&](unsigned long long w, uint32_t h) {
        uint32_t p = slot_of(h);
        for (int it = 0; it < TABLE; it++) {
            const unsigned long long old = __hip_atomic_fetch_min(&s_tab[p], w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (old == ~0ull || (uint32_t)(old >> 32) == (uint32_t)(w >> 32)) return;    // placed, or merged with the same vertex
            if (old > w) w = old;                                                          // displaced a larger word: carry it on
            p = (p + 1) & (TABLE - 1);
        }
        s_full = 1;
    };

    // the batch's vertices before this hop are the seeds (sampled_ids) and the nodes earlier hops added (the bucket's known
    // list -- or sampled_ids too when there is no list or it outgrew its capacity)
    const bool listed = a.known_pairs != nullptr && n_listed <= a.known_cap;
    const int32_t n_scan = listed ? n_seed : n_known;
    const LG_G unsigned long long* klist = a.known_pairs + (int64_t)b * a.known_cap;
#pragma unroll
    for (int u = 0; u < CL; u++)
        if (u * LG_DEDUP_THREADS + tid >= total) rp[u] = ~0ull;
    if (!listed || tid >= n_listed) kl0 = ~0ull;
    // passes: distinct vertices <= known + claims; keep the expected load of a pass at or below LG_LDS_FILL_16THS / 16 of the
    // table.  The bucket's share of the scanned ids is ESTIMATED (an even spread + a quarter; counting it would cost every
    // workgroup one more round trip to memory), and the hash is assumed to spread the bucket evenly over its sub-buckets: when
    // either is wrong (s_full: an insert found no free word) the whole bucket is redone with twice the passes -- every claim's
    // outcome is the same under any partition, so what finished passes already wrote is simply written again.
    const int32_t known_est = (listed ? n_listed : 0) + n_scan / NB + n_scan / (4 * NB) + 32;
    int32_t passes = 1;
    while ((int64_t)known_est + total > (int64_t)passes * (TABLE / 16 * LG_LDS_FILL_16THS)) passes <<= 1;
    // A bucket of at most CL claims per thread (the usual case) works from the registers.  A larger one is read
    // again, sweep by sweep; and a bucket whose list could not take all its claims (its count says so) reads the hop's slots
    // instead and keeps what hashes into this bucket.
    const bool from_slots = total > a.claim_cap;
    const bool resident = !from_slots && total <= CL * LG_DEDUP_THREADS;
    const int32_t n_src = from_slots ? hop_geometry(a).total : total;
    auto fetch = [&](int32_t k0, unsigned long long (&pr)[CL]) {
#pragma unroll
        for (int u = 0; u < CL; u++) {
            const int32_t k = k0 + u * LG_DEDUP_THREADS + tid;
            pr[u] = ~0ull;
            if (k >= n_src) continue;
            if (from_slots) {
                int32_t d = a.slot_dst[k];
                if (d == -1) continue;
                if (d < -1) d = LG_SLOT_LOSER(d);                // (an earlier pass may have marked the slot)
                if ((lg_tab_hash(d) & (NB - 1)) == (uint32_t)b) pr[u] = ((unsigned long long)(uint32_t)d << 32) | (uint32_t)k;
            } else {
                pr[u] = a.claim_pairs[lg_claim_at<NB>(b, k)];
            }
        }
    };

    for (;;) {
        const uint32_t pmask = (uint32_t)passes - 1u;
        bool overflow = false;
        for (uint32_t pass = 0; pass <= pmask; pass++) {
            for (int32_t i = tid; i < TABLE; i += LG_DEDUP_THREADS) s_tab[i] = ~0ull;
            if (tid == 0) s_full = 0;
            lds_barrier();
            for (int32_t i0 = 0; i0 < n_scan; i0 += LG_DEDUP_THREADS) {
                const int32_t i = i0 + tid;
                const int32_t id = i >= n_scan ? -1 : (i0 == 0 ? kid0 : a.sampled_ids[i]);      // (the first one came early)
                if (id < 0) continue;
                const uint32_t h = lg_tab_hash(id);
                if ((h & (NB - 1)) != (uint32_t)b || ((h >> BB) & pmask) != pass) continue;
                insert(((unsigned long long)(uint32_t)id << 32) | (uint32_t)i, h);
            }
            if (listed)
                for (int32_t i0 = 0; i0 < n_listed; i0 += LG_DEDUP_THREADS) {
                    const int32_t i = i0 + tid;
                    const unsigned long long pr = i0 == 0 ? kl0 : (i < n_listed ? klist[i] : ~0ull);
                    if (pr == ~0ull) continue;
                    const uint32_t h = lg_tab_hash((int32_t)(pr >> 32));
                    if (((h >> BB) & pmask) != pass) continue;
                    insert(pr, h);
                }
            for (int32_t k0 = 0; k0 < n_src; k0 += CL * LG_DEDUP_THREADS) {
                unsigned long long pr[CL];
                if (resident) {
#pragma unroll
                    for (int u = 0; u < CL; u++) pr[u] = rp[u];
                } else
                    fetch(k0, pr);
#pragma unroll
                for (int u = 0; u < CL; u++) {
                    if (pr[u] == ~0ull) continue;
                    const uint32_t h = lg_tab_hash((int32_t)(pr[u] >> 32));
                    if (((h >> BB) & pmask) != pass) continue;
                    insert((pr[u] & 0xFFFFFFFF00000000ull) | PENDING | (uint32_t)pr[u], h);
                }
            }
            lds_barrier();
            // (uniform: every thread reads s_full behind the barrier above, and nobody resets it before the next barrier every
            // thread passes -- the one below, or the one behind `passes <<= 1`)
            if (s_full != 0) { overflow = true; break; }
            for (int32_t k0 = 0; k0 < n_src; k0 += CL * LG_DEDUP_THREADS) {
                unsigned long long pr[CL];
                if (resident) {
#pragma unroll
                    for (int u = 0; u < CL; u++) pr[u] = rp[u];
                } else
                    fetch(k0, pr);
#pragma unroll
                for (int u = 0; u < CL; u++) {
                    if (pr[u] == ~0ull) continue;
                    const uint32_t id = (uint32_t)(pr[u] >> 32), slot = (uint32_t)pr[u];
                    const uint32_t h = lg_tab_hash((int32_t)id);
                    if (((h >> BB) & pmask) != pass) continue;
                    uint32_t p = slot_of(h);
                    uint32_t v = 0xFFFFFFFFu;
                    for (int it = 0; it < TABLE; it++) {
                        const unsigned long long w = s_tab[p];
                        if ((uint32_t)(w >> 32) == id) { v = (uint32_t)w; break; }
                        p = (p + 1) & (TABLE - 1);
                    }
                    if (v != (PENDING | slot)) {          // not the lowest slot of a new vertex
                        a.slot_dst[slot] = LG_SLOT_LOSER((int32_t)id);
                        a.slot_pos[slot] = (v & PENDING) ? -2 - (int32_t)(v & ~PENDING) : (int32_t)v;
                    }
                }
            }
            lds_barrier();                                 // (the next pass clears the table)
        }
        if (!overflow) break;
        if (passes >= (1 << 14)) {                             // 2^14 sub-buckets of one bucket still too full: not a hash problem
            if (tid == 0) raise_error(a.hop_scratch, a.err_flag, LG_ERR_TABLE_FULL);
            break;
        }
        passes <<= 1;
        lds_barrier();      // every thread has read s_full before thread 0 clears it for the retry (without this a fast wave
                            // could reset it while a slow one had not yet left the barrier above: a split workgroup)
    }
    if (tid == 0) a.claim_cnt[b * LG_CLAIM_CNT_STRIDE] = 0;      // the next hop's sampling starts an empty list
}

// ------------------------------------------------------------------------------------------
// K2: compaction in ONE pass over the hop's slots (rounds 1-2 took three: per-tile counts, a one-workgroup prefix with
// counter_update, the scatter).  A workgroup takes the next 1024-slot super tile by ticket, counts its valid edges and first
// touches (wave ballots: the sample / de-duplication kernels left a mark on every slot that lost its first touch), and gets
// what the earlier super tiles hold by decoupled look-back over one 64-bit word per super tile
//     [ status : 2 | edges : 31 | nodes : 31 ]   status 1 = this tile's own counts, 2 = inclusive prefix
// (tickets are handed out in order, so every tile a workgroup waits for belongs to a workgroup that is already running).
// Then the slot-ordered compaction itself: ballot + mbcnt prefix inside the tile -> edges (global ids + both local
// positions when known), new nodes and, next to every edge, the row header of the sampled neighbour, so that the next
// hop's frontier needs no dependent lookup.
// lds form: a slot that lost to ANOTHER slot of the hop needs that winner's new position.  sample_kernel left -1 in slot_pos
// of every slot and the de-duplication kernel wrote the losers' entries only, so a winner's entry still holds -1 when this
// kernel starts: the winner publishes its position there and the loser -- whose winner is always the LOWER slot, in this
// super tile or an earlier one, so its position is on its way -- polls that one word.  No pass over the edges afterwards.
// Everything that crosses workgroups here (status words, published positions, tickets) is a single self-contained word moved
// with relaxed agent-scope atomics: no acquire / release fences, which on this part write back and invalidate a whole XCD's
// L2 (measured: with fences the kernel took 1.1 ms per 256-lane group instead of ~0.1).
// The workgroup that finishes last does what counter_update(op) does (operator_impl.cu:69-82), leaves the hop scratch for
// the kernels that follow, and zeroes the status words and tickets for the next hop.
// ------------------------------------------------------------------------------------------
#ifndef LG_SCATTER_MIN_WAVES
#define LG_SCATTER_MIN_WAVES 5
#endif
#ifndef LG_SCATTER_MIN_WAVES_LAST
#define LG_SCATTER_MIN_WAVES_LAST 8
#endif
#ifndef LG_COMPACT_THREADS
#define LG_COMPACT_THREADS 256
#endif
#define LG_SPIN_LIMIT (1 << 24)       // polls of one word before a waiter gives up with LG_ERR_CHAIN (seconds; a wait is microseconds)
#define LG_ST_AGG (1ull << 62)
#define LG_ST_PREF (2ull << 62)
__device__ __forceinline__ unsigned long long st_word(unsigned long long status, int32_t e, int32_t n)
{
    return status | ((unsigned long long)(uint32_t)e << 31) | (unsigned long long)(uint32_t)n;
}
__device__ __forceinline__ int32_t st_edges(unsigned long long w) { return (int32_t)((w >> 31) & 0x7FFFFFFFull); }
__device__ __forceinline__ int32_t st_nodes(unsigned long long w) { return (int32_t)(w & 0x7FFFFFFFull); }

template <bool LAST, int CT>       // LAST: the last hop writes no frontier headers and no position state (fewer registers, more waves per SIMD); CT: threads
                                   // per workgroup = slots per 'tile row' (a workgroup iteration takes LG_SLOTS_PER_LANE * CT consecutive slots)
__global__ __launch_bounds__(CT, LAST ? LG_SCATTER_MIN_WAVES_LAST : LG_SCATTER_MIN_WAVES) __attribute__((amdgpu_num_sgpr(80)))
void compact_kernel(HopParams hp, const LanePtrs* __restrict__ lanes)
{
    constexpr int NW = LG_SLOTS_PER_LANE * (CT / 64);     // waves' worth of slots in a workgroup iteration (16 or 32)
    constexpr int CSUPER = LG_SLOTS_PER_LANE * CT;
    static_assert(NW <= 64, "one lane of wave 0 per (u, wave)");
    const SampleArgs a = lane_args(hp, lanes);
    __shared__ int32_t s_cnt[2][NW];               // [valid | first touch][u * 4 + wave]
    __shared__ int32_t s_pre[2][NW];               // exclusive prefix of s_cnt inside the super tile
    __shared__ unsigned long long s_mf[NW];        // first-touch ballots
    __shared__ int32_t s_st, s_ex[2], s_last, s_tot[2];
    const HopGeom g = hop_geometry(a);
    LG_G int32_t* hs = a.hop_scratch;
    LG_G int32_t* nc = a.node_counter;
    LG_G int32_t* ec = a.edge_counter;
    const int32_t nc0 = nc[0], nc1 = nc[1], ec0 = ec[0], ec1 = ec[1];     // (rewritten by the LAST workgroup only)
    const int32_t total = g.total, nsuper = (g.total + CSUPER - 1) / CSUPER;
    const bool seeds = a.op_id == INTRABATCH_CON;
    const int32_t f_off = g.frontier_off;
    const int32_t node_base = nc0 + nc1, edge_base = ec0 + ec1;           // operator_impl.cu:268, :275
    const LG_G int32_t* frontier = g.frontier;
    LG_G unsigned long long* state = a.tile_state;
    const int32_t tid = threadIdx.x;
    const int32_t wave = tid >> 6, lane = tid & 63;
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));

    // The next tile's ticket is drawn BEFORE this tile's stores are issued and the barriers inside the loop order LDS only: memory
    // operations of a wave complete in order, so a ticket drawn after the stores (and a barrier that fences global memory) waits
    // for the whole tile to have been written before the next tile's loads can even be addressed.
    if (tid == 0) s_st = __hip_atomic_fetch_add(hs + HS_CTICKET, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    for (;;) {
        const int32_t st = s_st;
        if (st >= nsuper) break;                                          // (uniform)
        const int32_t idx0 = st * CSUPER;
        int32_t v[LG_SLOTS_PER_LANE];
        unsigned long long mv[LG_SLOTS_PER_LANE], mf[LG_SLOTS_PER_LANE];
#pragma unroll
        for (int u = 0; u < LG_SLOTS_PER_LANE; u++) {
            const int32_t idx = idx0 + u * CT + tid;
            v[u] = idx < total ? a.slot_dst[idx] : -1;
        }
        // what depends on the slot INDEX only -- the vertex the slot sampled for, its position, the carried cache slot -- is loaded
        // together with slot_dst, for every slot of the tile: more bytes (invalid slots too), one dependent round trip less (-4 %, round 4)
        int32_t src_of[LG_SLOTS_PER_LANE], src_pos[LG_SLOTS_PER_LANE], fsv[LG_SLOTS_PER_LANE];
#pragma unroll
        for (int u = 0; u < LG_SLOTS_PER_LANE; u++) {
            const int32_t idx = idx0 + u * CT + tid;
            src_of[u] = 0; src_pos[u] = 0; fsv[u] = LG_FS_UNKNOWN;
            if (idx < total) {
                const int32_t q = idx / a.count;
                src_of[u] = frontier[q];
                src_pos[u] = seeds ? q : a.agg_src_off[f_off + q];
                if (a.slot_fs != nullptr) fsv[u] = a.slot_fs[idx];
            }
        }
#pragma unroll
        for (int u = 0; u < LG_SLOTS_PER_LANE; u++) {
            const bool valid = v[u] != -1;
            const bool first = v[u] >= 0;                     // the de-duplication did not mark it a loser (LG_SLOT_LOSER)
            if (v[u] < -1) v[u] = LG_SLOT_LOSER(v[u]);        // valid slots hold the vertex id from here on; invalid ones -1
            mv[u] = __ballot(valid);
            mf[u] = __ballot(first);
            if (lane == 0) {
                s_cnt[0][u * (CT / 64) + wave] = __popcll(mv[u]);
                s_cnt[1][u * (CT / 64) + wave] = __popcll(mf[u]);
                s_mf[u * (CT / 64) + wave] = mf[u];
            }
        }
        lds_barrier();
        // wave 0: this super tile's counts, for everybody behind it
        int32_t te = 0, tn = 0;
        if (wave == 0) {
            const int32_t ce = lane < NW ? s_cnt[0][lane] : 0, cn = lane < NW ? s_cnt[1][lane] : 0;
            int32_t ie = ce, in = cn;
            for (int d = 1; d < NW; d <<= 1) {
                const int32_t oe = __shfl_up(ie, d), on = __shfl_up(in, d);
                if (lane >= d) { ie += oe; in += on; }
            }
            te = __shfl(ie, NW - 1);
            tn = __shfl(in, NW - 1);
            if (lane < NW) {
                s_pre[0][lane] = ie - ce;
                s_pre[1][lane] = in - cn;
            }
            if (st > 0 && lane == 0)
                __hip_atomic_store(state + st, st_word(LG_ST_AGG, te, tn), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // phase 1: every load of the thread's four slots that does not need the prefix -- in flight while wave 0 looks back
        // (nothing is stored in between: the buffers may alias as far as the compiler knows)
        int32_t lost_pos[LG_SLOTS_PER_LANE];
        RowHdr nh[LG_SLOTS_PER_LANE];
#pragma unroll
        for (int u = 0; u < LG_SLOTS_PER_LANE; u++) {
            const int32_t idx = idx0 + u * CT + tid;
            if (v[u] >= 0) {
                const bool first = (mf[u] >> lane) & 1ull;
                if (!first) fsv[u] = LG_FS_UNKNOWN;
                if (!LAST) nh[u] = load_hdr(a.row_hdr + v[u]);     // next hop's frontier header
                lost_pos[u] = first ? 0 : a.slot_pos[idx];         // final already, or -2 - (slot it lost to)
            }
        }
        if (wave == 0) {
            int32_t xe = 0, xn = 0;
            if (st > 0) {
                int32_t j = st - 1;                                       // nearest earlier tile not yet accounted for
                for (;;) {
                    const int32_t me = j - lane;
                    unsigned long long w = LG_ST_PREF;                    // below tile 0: an inclusive prefix of nothing
                    if (me >= 0) {
                        w = __hip_atomic_load(state + me, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        for (int32_t spin = 0; (w >> 62) == 0; spin++) {
                            if (spin == LG_SPIN_LIMIT) {       // cannot happen (the tile's workgroup is running): give up, never hang
                                raise_error(a.hop_scratch, a.err_flag, LG_ERR_CHAIN);
                                w = LG_ST_AGG;
                                break;
                            }
                            __builtin_amdgcn_s_sleep(1);
                            w = __hip_atomic_load(state + me, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    }
                    const unsigned long long pm = __ballot((w >> 62) == 2);
                    const int first_p = pm ? __builtin_ctzll(pm) : 64;   // lanes up to the nearest inclusive prefix count
                    if (lane <= first_p) { xe += st_edges(w); xn += st_nodes(w); }
                    if (pm) break;
                    j -= 64;
                }
                for (int off = 32; off > 0; off >>= 1) { xe += __shfl_down(xe, off); xn += __shfl_down(xn, off); }
                xe = __shfl(xe, 0);
                xn = __shfl(xn, 0);
            }
            if (lane == 0) {
                s_ex[0] = xe;
                s_ex[1] = xn;
                __hip_atomic_store(state + st, st_word(LG_ST_PREF, xe + te, xn + tn), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        lds_barrier();
        const int32_t xe = s_ex[0], xn = s_ex[1];
        // the first touches' positions are known now -- publish them before anything else, later super tiles' losers
        // are waiting for nothing but this (publishing with the other stores below would chain every tile's loads behind the
        // stores of the tiles before it)
        int32_t n_at[LG_SLOTS_PER_LANE];
#pragma unroll
        for (int u = 0; u < LG_SLOTS_PER_LANE; u++) {
            n_at[u] = -1;
            if ((mf[u] >> lane) & 1ull) {
                n_at[u] = node_base + xn + s_pre[1][u * (CT / 64) + wave] + __popcll(mf[u] & lt);
                __hip_atomic_store(a.slot_pos + idx0 + u * CT + tid, n_at[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        {
            // a slot that lost to ANOTHER slot of the hop: that slot IS the winner (chains have length one), and it is a LOWER slot
#pragma unroll
            for (int u = 0; u < LG_SLOTS_PER_LANE; u++) {
                if (v[u] < 0 || lost_pos[u] >= -1) continue;
                const int32_t w = -2 - lost_pos[u];
                if (w >= idx0) {            // of this super tile: its position follows from the ballots at hand
                    const int32_t ww = (w - idx0) >> 6, lw = w & 63;
                    lost_pos[u] = node_base + xn + s_pre[1][ww] + __popcll(s_mf[ww] & (lw == 0 ? 0ull : (~0ull >> (64 - lw))));
                } else {                    // of an earlier one: wait for the position its workgroup publishes
                    const LG_G int32_t* wp = a.slot_pos + w;
                    int32_t np = __hip_atomic_load(wp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    for (int32_t spin = 0; np < 0; spin++) {
                        if (spin == LG_SPIN_LIMIT) {           // (as above: an error bit instead of a hung GPU)
                            raise_error(a.hop_scratch, a.err_flag, LG_ERR_CHAIN);
                            np = 0;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(1);
                        np = __hip_atomic_load(wp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    lost_pos[u] = np;
                }
            }
        }
        // phase 2: the stores (V2: behind the draw of the next ticket)
        int32_t next_st = 0;
        if (tid == 0) next_st = __hip_atomic_fetch_add(hs + HS_CTICKET, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int u = 0; u < LG_SLOTS_PER_LANE; u++) {
            if (v[u] < 0) continue;
            const int32_t dst = v[u];
            const int32_t e = edge_base + xe + s_pre[0][u * (CT / 64) + wave] + __popcll(mv[u] & lt);
#ifdef LG_COMPACT_NO_NT
            constexpr bool NT = false;
#else
            constexpr bool NT = LAST;
#endif
            if (NT) {       // nobody on this GPU reads the last hop's edge arrays again: non-temporal stores (chain -5..-19 us and the gathers
                              // behind it -25..-50 us on one box: what they read -- sampled_ids, node_slot -- stays cached)
                __builtin_nontemporal_store(dst, &a.agg_src_ids[e]);
                __builtin_nontemporal_store(src_of[u], &a.agg_dst_ids[e]);
                __builtin_nontemporal_store(src_pos[u], &a.agg_dst_off[e]);
            } else {
                a.agg_src_ids[e] = dst;                            // :256, :276
                a.agg_dst_ids[e] = src_of[u];                      // :257, :277
                a.agg_dst_off[e] = src_pos[u];
            }
            if (!LAST) store_hdr(a.fh_edge + e, nh[u]);
            const int32_t n = n_at[u];
            if (n >= 0) {
                a.sampled_ids[n] = dst;                            // :270
                if (a.node_slot != nullptr) a.node_slot[n] = fsv[u];
                // (:271 -- no position map: later hops find the node in its bucket's known list, list_known_kernel)
                if (NT) __builtin_nontemporal_store(n, &a.agg_src_off[e]);
                else a.agg_src_off[e] = n;                              // construct_graph's neighbour side, known here
            } else {
                if (NT) __builtin_nontemporal_store(lost_pos[u], &a.agg_src_off[e]);
                else a.agg_src_off[e] = lost_pos[u];
            }
        }
        if (tid == 0) s_st = next_st;
        lds_barrier();
    }

    // the workgroup that finishes last: counter_update(op_id), op_id % 3 == 0 (operator_impl.cu:69-82), with nc[6] = n_new and
    // ec[2] = n_edge being what the reference's atomicAdds (:263-264) leave there; hop scratch; status words back to zero
    if (tid == 0) {
        // this thread wrote the tiles' status words: they must have landed before the count says "finished", or the last
        // workgroup could read state[nsuper-1] before its PREF word is there, or its zeroes could be overtaken by one of them.
        // A workgroup-scope release fence does NOT wait for the thread's own global stores on gfx950 (it compiles to nothing
        // before the atomic: global_store ... ; global_atomic_add); what orders them is an explicit wait for this thread's
        // outstanding memory operations -- vmcnt(0): stores count in vmcnt on CDNA -- and nothing else is needed: the status
        // words are self-contained and read with agent-scope atomics at the L2 they were written through.
        __builtin_amdgcn_s_waitcnt(0x0F70);                // vmcnt(0), expcnt / lgkmcnt untouched
        s_last = (__hip_atomic_fetch_add(hs + HS_CDONE, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int32_t)gridDim.x - 1) ? 1 : 0;
    }
    __syncthreads();
    if (!s_last) return;
    if (tid == 0) {
        const unsigned long long tw = nsuper > 0 ? __hip_atomic_load(state + nsuper - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
        s_tot[0] = st_edges(tw);
        s_tot[1] = st_nodes(tw);
    }
    __syncthreads();
    for (int32_t t = tid; t < nsuper; t += CT) state[t] = 0ull;
    if (tid == 0) {
        const int32_t n_edge = s_tot[0], n_new = s_tot[1];
        hs[HS_CTICKET] = 0;
        hs[HS_CDONE] = 0;
        hs[HS_FRONTIER_IS_SEEDS] = seeds ? 1 : 0;
        hs[HS_FRONTIER_OFF] = f_off;
        hs[HS_FRONTIER_LEN] = g.frontier_len;
        hs[HS_NODE_BASE] = node_base;
        hs[HS_EDGE_BASE] = edge_base;
        hs[HS_N_NEW] = n_new;
        hs[HS_N_EDGE] = n_edge;
        hs[HS_SLOTS] = total;
        const int32_t h = a.op_id / INTRABATCH_CON;
        hs[HS_RANGE + 2 * h] = node_base;                  // range snapshot for this hop's gather
        hs[HS_RANGE + 2 * h + 1] = n_new;
        nc[0] = node_base;
        nc[1] = n_new;
        nc[INTRABATCH_CON * 2] = 0;
        nc[INTRABATCH_CON * 2 + 1] = node_base + n_new;
        ec[0] = edge_base;
        ec[1] = n_edge;
        ec[2] = 0;
        nc[INTRABATCH_CON * 3 + h] = node_base + n_new;
        ec[INTRABATCH_CON * 3 + h] = edge_base + n_edge;
    }
}

// ------------------------------------------------------------------------------------------
// K4b (lds form, every hop but the last): the nodes this hop added, appended as (vertex << 32 | position) to the lane's
// per-bucket lists -- the next hops' de-duplication workgroups read only their bucket's list.  A chunk of new nodes is
// counted per bucket in LDS, ONE global atomicAdd per non-empty bucket reserves its entries, a second sweep places them.
// A list that outgrows its capacity is not used (its count says so; that bucket's workgroup scans sampled_ids instead).
// ------------------------------------------------------------------------------------------
#define LG_LIST_CHUNK 8192
template <int BB>
__global__ __launch_bounds__(LG_TILE) void list_known_kernel(HopParams hp, const LanePtrs* __restrict__ lanes)
{
    constexpr int NB = 1 << BB;
    const SampleArgs a = lane_args(hp, lanes);
    __shared__ int32_t s_kcnt[NB], s_kbase[NB];
    if (a.known_pairs == nullptr) return;
    const int32_t tid = threadIdx.x;
    const int32_t h = a.op_id / INTRABATCH_CON;
    const int32_t base = a.hop_scratch[HS_RANGE + 2 * h], n_new = a.hop_scratch[HS_RANGE + 2 * h + 1];   // this hop's range of sampled_ids
    for (int32_t c0 = blockIdx.x * LG_LIST_CHUNK; c0 < n_new; c0 += gridDim.x * LG_LIST_CHUNK) {
        const int32_t c1 = min(c0 + LG_LIST_CHUNK, n_new);
        for (int32_t i = tid; i < NB; i += LG_TILE) s_kcnt[i] = 0;
        __syncthreads();
        for (int32_t i = c0 + tid; i < c1; i += LG_TILE)
            atomicAdd(&s_kcnt[lg_tab_hash(a.sampled_ids[base + i]) & (NB - 1)], 1);
        __syncthreads();
        for (int32_t i = tid; i < NB; i += LG_TILE) {
            const int32_t c = s_kcnt[i];
            s_kbase[i] = c ? __hip_atomic_fetch_add(a.known_cnt + i, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
            s_kcnt[i] = 0;
        }
        __syncthreads();
        for (int32_t i = c0 + tid; i < c1; i += LG_TILE) {
            const int32_t id = a.sampled_ids[base + i];
            const int32_t bk = (int32_t)(lg_tab_hash(id) & (NB - 1));
            const int32_t at = s_kbase[bk] + atomicAdd(&s_kcnt[bk], 1);
            if (at < a.known_cap)
                a.known_pairs[(int64_t)bk * a.known_cap + at] = ((unsigned long long)(uint32_t)id << 32) | (uint32_t)(base + i);
        }
        __syncthreads();
    }
}

void launch_random_sample(hipStream_t s, const HopParams& p, const LanePtrs* d_lanes, int32_t n_lanes)
{
    // Fixed grids that stride over super tiles; grid.y = lanes (independent mini-batches of a group).
    int32_t max_super = (p.max_slots + LG_SUPER - 1) / LG_SUPER;
    if (max_super < 1) max_super = 1;
    int32_t gx = max_super < 1024 ? max_super : 1024;
    const int max_wg = tuning().sample_max_wg;
    while (gx > 64 && (int64_t)gx * n_lanes > max_wg) gx /= 2;  // keep the whole launch near 2 x resident capacity
    while (gx > 1 && (int64_t)gx * n_lanes > max_wg && max_wg < 4096) gx /= 2;   // (experiments with fewer workgroups)
    {
        // Equal workgroups that fill the machine about twice leave its second round half empty: between one and six rounds' worth
        // (8 workgroups of this kernel per CU x 256 CUs), take ONE round of longer-lived workgroups instead.  Measured (one_round_ab.txt):
        // 64 lanes at B = 8000 (3 904 -> 2 048 workgroups) +1 %, 128 lanes at B = 4096 +1.4 %, 256 lanes and D = 256 the same within
        // the noise; 512 lanes at B = 1024 (15 rounds) are not touched.
        const int64_t resident = 8 * 256, total = (int64_t)gx * n_lanes;
        if (total > resident && total < 6 * resident) gx = (int32_t)std::max<int64_t>(1, resident / n_lanes);
    }
    const dim3 grid(gx, n_lanes);
    {
        // buckets per lane follow the pool's largest hop (legion_core.h).  8 / 16 / 64 buckets: the sampling kernel writes the claim
        // lists itself (64: staged per super tile in LDS).  256 buckets: it samples partition tiles of K super tiles and place_kernel
        // writes the lists; K follows THIS hop: as large as the staging allows (LG_PLACE_MAX_K) unless that leaves the launch with
        // fewer than ~8 k workgroups by the hop's capacity (a hop typically fills a quarter of it: ~2 k active ones; measured at
        // B = 8000: 2 k -> 8 k +1...2 %, beyond: the same), and never below the class's minimum (run_off is sized by it, storage.hip)
        HopParams q = p;
        const bool small = p.lds_bucket_bits == LG_LDS_BITS_SMALL || p.lds_bucket_bits == LG_LDS_BITS_SMALL16;
        const int32_t k_lo = lg_lds_k_min(p.lds_bucket_bits);
        int32_t k = (small || p.lds_bucket_bits == LG_LDS_BITS_MEDIUM) ? 1 : LG_PLACE_MAX_K;
        const int want_wg = tuning().lds_part_wg;
        while (k > k_lo && (int64_t)(max_super / k) * n_lanes < want_wg) k /= 2;
        q.lds_k = k;
        int32_t gp = (max_super + k - 1) / k;                  // one workgroup per partition tile ...
        while (gp > 16 && (int64_t)gp * n_lanes > 16384) gp = (gp + 1) / 2;  // ... within reason
        const dim3 pgrid(gp, n_lanes);
        const size_t stage = (size_t)k * LG_SUPER * sizeof(unsigned long long);
        switch (p.lds_bucket_bits) {
        case LG_LDS_BITS_SMALL:
            sample_kernel<LG_LDS_BITS_SMALL, true><<<grid, LG_TILE, 0, s>>>(q, d_lanes);
            hipCheckError();
            dedup_lists_kernel<LG_LDS_BITS_SMALL, LG_DEDUP_CLAIMS><<<dim3(1 << LG_LDS_BITS_SMALL, n_lanes), LG_DEDUP_THREADS, 0, s>>>(q, d_lanes);
            break;
        case LG_LDS_BITS_SMALL16:
            sample_kernel<LG_LDS_BITS_SMALL16, true><<<grid, LG_TILE, 0, s>>>(q, d_lanes);
            hipCheckError();
            dedup_lists_kernel<LG_LDS_BITS_SMALL16, LG_DEDUP_CLAIMS><<<dim3(1 << LG_LDS_BITS_SMALL16, n_lanes), LG_DEDUP_THREADS, 0, s>>>(q, d_lanes);
            break;
        case LG_LDS_BITS_MEDIUM:
            sample_kernel<LG_LDS_BITS_MEDIUM, true, true><<<grid, LG_TILE, 0, s>>>(q, d_lanes);
            hipCheckError();
            // (a bucket of up to CL x 1024 claims is worked on from registers, whatever the number of passes over its sub-buckets)
            if (p.dedup_claims == LG_DEDUP_CLAIMS_BIG) dedup_lists_kernel<LG_LDS_BITS_MEDIUM, LG_DEDUP_CLAIMS_BIG, LG_DEDUP_BIG_TABLE_BITS><<<dim3(1 << LG_LDS_BITS_MEDIUM, n_lanes), LG_DEDUP_THREADS, 0, s>>>(q, d_lanes);
            else if (p.dedup_claims == LG_DEDUP_CLAIMS_MID) dedup_lists_kernel<LG_LDS_BITS_MEDIUM, LG_DEDUP_CLAIMS_MID><<<dim3(1 << LG_LDS_BITS_MEDIUM, n_lanes), LG_DEDUP_THREADS, 0, s>>>(q, d_lanes);
            else dedup_lists_kernel<LG_LDS_BITS_MEDIUM, LG_DEDUP_CLAIMS><<<dim3(1 << LG_LDS_BITS_MEDIUM, n_lanes), LG_DEDUP_THREADS, 0, s>>>(q, d_lanes);
            break;
        default:
            sample_kernel<LG_LDS_BITS_LARGE, false><<<pgrid, LG_TILE, 0, s>>>(q, d_lanes);
            hipCheckError();
            place_kernel<LG_LDS_BITS_LARGE><<<pgrid, LG_PLACE_THREADS, stage, s>>>(q, d_lanes);
            hipCheckError();
            if (p.dedup_claims == LG_DEDUP_CLAIMS_MID) dedup_lists_kernel<LG_LDS_BITS_LARGE, LG_DEDUP_CLAIMS_MID><<<dim3(1 << LG_LDS_BITS_LARGE, n_lanes), LG_DEDUP_THREADS, 0, s>>>(q, d_lanes);
            else dedup_lists_kernel<LG_LDS_BITS_LARGE, LG_DEDUP_CLAIMS><<<dim3(1 << LG_LDS_BITS_LARGE, n_lanes), LG_DEDUP_THREADS, 0, s>>>(q, d_lanes);
            break;
        }
    }
    hipCheckError();
    // compaction: LG_COMPACT_THREADS per workgroup (a workgroup iteration takes 4 x that many consecutive slots), as many workgroups per
    // lane as the sampling launch has per 1024 slots' worth
    {
        const dim3 cgrid(std::max(1, (int)grid.x * LG_TILE / LG_COMPACT_THREADS), n_lanes);
        if (p.last_hop) compact_kernel<true, LG_COMPACT_THREADS><<<cgrid, LG_COMPACT_THREADS, 0, s>>>(p, d_lanes);
        else compact_kernel<false, LG_COMPACT_THREADS><<<cgrid, LG_COMPACT_THREADS, 0, s>>>(p, d_lanes);
    }
    hipCheckError();
    if (!p.last_hop) {       // later hops must recognise the nodes this one added: their buckets' lists
        int32_t chunks = (p.max_slots + LG_LIST_CHUNK - 1) / LG_LIST_CHUNK;
        if (chunks > 256) chunks = 256;
        if (p.lds_bucket_bits == LG_LDS_BITS_SMALL) list_known_kernel<LG_LDS_BITS_SMALL><<<dim3(chunks, n_lanes), LG_TILE, 0, s>>>(p, d_lanes);
        else if (p.lds_bucket_bits == LG_LDS_BITS_SMALL16) list_known_kernel<LG_LDS_BITS_SMALL16><<<dim3(chunks, n_lanes), LG_TILE, 0, s>>>(p, d_lanes);
        else if (p.lds_bucket_bits == LG_LDS_BITS_MEDIUM) list_known_kernel<LG_LDS_BITS_MEDIUM><<<dim3(chunks, n_lanes), LG_TILE, 0, s>>>(p, d_lanes);
        else list_known_kernel<LG_LDS_BITS_LARGE><<<dim3(chunks, n_lanes), LG_TILE, 0, s>>>(p, d_lanes);
        hipCheckError();
    }
}

// ------------------------------------------------------------------------------------------
// per-vertex row headers (GraphStorage): every vertex starts in the full CSR (slot P) ...
// ------------------------------------------------------------------------------------------
__global__ void init_row_hdr_kernel(RowHdr* __restrict__ hdr, const int64_t* __restrict__ csr_index, int32_t n,
                                    int32_t slot)
{
    for (int32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < n; v += gridDim.x * blockDim.x) {
        RowHdr h;
        h.start = csr_index[v];
        h.deg = (int32_t)(csr_index[v + 1] - h.start);
        h.slot = slot;
        hdr[v] = h;
    }
}

void init_row_headers(hipStream_t s, RowHdr* hdr, const int64_t* csr_index, int32_t n, int32_t slot)
{
    int32_t grid = (n + 255) / 256;
    if (grid > 4096) grid = 4096;
    if (grid < 1) grid = 1;
    init_row_hdr_kernel<<<grid, 256, 0, s>>>(hdr, csr_index, n, slot);
    hipCheckError();
}

// ... and the vertices GPU `Ki` of the clique caches (QT[r*Kg + Ki], r < capacity) point into its CSR
__global__ void cache_row_hdr_kernel(RowHdr* __restrict__ hdr, const int32_t* __restrict__ QT, int32_t Kg, int32_t Ki,
                                     int32_t capacity, int32_t n, const int64_t* __restrict__ d_index, int32_t slot)
{
    for (int32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < capacity; r += gridDim.x * blockDim.x) {
        const int64_t t = (int64_t)r * Kg + Ki;
        if (t >= n) continue;
        RowHdr h;
        h.start = d_index[r];
        h.deg = (int32_t)(d_index[r + 1] - h.start);
        h.slot = slot;
        hdr[QT[t]] = h;
    }
}

void cache_row_headers(hipStream_t s, RowHdr* hdr, const int32_t* QT, int32_t Kg, int32_t Ki, int32_t capacity,
                       int32_t n, const int64_t* d_index, int32_t slot)
{
    if (capacity <= 0) return;
    int32_t grid = (capacity + 255) / 256;
    if (grid > 4096) grid = 4096;
    cache_row_hdr_kernel<<<grid, 256, 0, s>>>(hdr, QT, Kg, Ki, capacity, n, d_index, slot);
    hipCheckError();
}

// ------------------------------------------------------------------------------------------
// end of batch (IOComplete).  The reference zeroes position_map for every node of the batch
// (ClearPosMap, operator_impl.cu:542-548) and memsets the N/8-byte bitmap at the next batch's
// start (:151).  Here there is nothing to clear: no per-vertex state exists.  What is left of the op: the batch's counters
// as the trainer end will read them, and the device-resident iteration used by graph replay.  One wave per lane.
// ------------------------------------------------------------------------------------------
__global__ void end_of_batch_kernel(const LanePtrs* __restrict__ lanes, int32_t* __restrict__ iter_state)
{
    const BracketLane L = bracket_lane(lanes[blockIdx.y]);
    // GPURunner's lanes: the batch's counters in host-visible memory, as the trainer end will read them -- node_counter[2..3]
    // already holding what the last gather op leaves there (counter_update(op%3==1), operator_impl.cu:83-85: the range of the
    // last hop's new nodes), whether or not that gather has run yet.  Visible to the host once the launch group has completed.
    if (L.counter_mirror != nullptr && threadIdx.x < 32) {
        const int32_t t = threadIdx.x;
        int32_t v = t < 16 ? L.node_counter[t] : L.edge_counter[t - 16];
        const int32_t hop_num = L.node_counter[INTRABATCH_CON * 3 - 1];
        if (t == 2 && hop_num >= 0 && hop_num <= 6) v = L.hop_scratch[HS_RANGE + 2 * hop_num];
        if (t == 3 && hop_num >= 0 && hop_num <= 6) v = L.hop_scratch[HS_RANGE + 2 * hop_num + 1];
        L.counter_mirror[t] = v;
    }
    // (every lane's batch_generate read the iteration in an earlier kernel of this stream)
    if (iter_state != nullptr && blockIdx.y == 0 && threadIdx.x == 0) iter_state[0] += iter_state[1];
}

void launch_end_of_batch(hipStream_t s, const LanePtrs* d_lanes, int32_t n_lanes, int32_t* iter_state)
{
    end_of_batch_kernel<<<dim3(1, n_lanes), 64, 0, s>>>(d_lanes, iter_state);
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
