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
//   * sample_kernel: a workgroup owns 1024 consecutive slots, four per lane; the tile's frontier row
//     headers (row start, degree, CSR slot) are staged once in LDS; the minstd draw is a table-driven
//     modular power + one IEEE double divide; the neighbour is published with
//     atomicMin(position_state[dst], PENDING + slot) so that the LOWEST slot owns a first touch
//     (deterministic, unlike atomicOr on a bitmap), and the value the atomic returns tells which slot
//     lost (slot_mark / slot_pos), so nothing re-reads the state array;
//   * dedup_lds_kernel (lds form of the first-touch state, the default): a hop's claims de-duplicated bucket by bucket in LDS;
//   * compact_kernel: ONE pass -- tiles handed out by ticket, counts of valid edges / first touches by wave ballots, the
//     prefix over a lane's tiles by decoupled look-back, then the slot-ordered compaction of edges (global ids + both local
//     positions) and of new nodes, + the next hop's row headers; its last workgroup does the counter_update state machine,
//     so there is no host round trip and no <<<1,1>>> launch;
//   * localise_kernel (atomics forms only): agg_src_off[e] of the edges whose neighbour another slot of the hop owns.
// Every kernel runs with grid.y = lanes (independent mini-batches, LanePtrs) and takes its pointers
// from the lane descriptor as global-address-space pointers.
// Every kernel is a fixed-size grid that strides over tiles and reads the frontier length from
// device memory, so the whole hop is enqueued without knowing any size on the host.
//
// Bound: the rate of scattered 4-byte atomics and loads (~19 G atomics/s on data-dependent addresses),
// not HBM bytes; no MFMA.
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
    LG_G int32_t* hop_scratch; LG_G uint32_t* position_map; LG_G int32_t* slot_mark; LG_G int32_t* node_slot;
    LG_G unsigned long long* pos_table; uint32_t pos_mask; LG_G int32_t* err_flag;
    LG_G int32_t* known_cnt; LG_G int32_t* claim_cnt; int32_t lds_buckets;
    LG_G int32_t* counter_mirror;
    int32_t total_num_nodes, max_slots;
};
__device__ __forceinline__ BracketLane bracket_lane(const LanePtrs& P)
{
    BracketLane L;
    L.sampled_ids = LG_GPTR(int32_t, P.sampled_ids); L.labels = LG_GPTR(int32_t, P.labels);
    L.node_counter = LG_GPTR(int32_t, P.node_counter); L.edge_counter = LG_GPTR(int32_t, P.edge_counter);
    L.hop_scratch = LG_GPTR(int32_t, P.hop_scratch); L.position_map = LG_GPTR(uint32_t, P.position_map);
    L.slot_mark = LG_GPTR(int32_t, P.slot_mark); L.node_slot = LG_GPTR(int32_t, P.node_slot);
    L.pos_table = LG_GPTR(unsigned long long, P.pos_table); L.pos_mask = P.pos_table_mask;
    L.err_flag = LG_GPTR(int32_t, P.err_flag);
    L.known_cnt = LG_GPTR(int32_t, P.known_cnt); L.claim_cnt = LG_GPTR(int32_t, P.claim_cnt); L.lds_buckets = P.lds_buckets;
    L.counter_mirror = LG_GPTR(int32_t, P.counter_mirror);
    L.total_num_nodes = P.total_num_nodes; L.max_slots = P.max_slots;
    return L;
}

__device__ __forceinline__ void raise_error(LG_G int32_t* hop_scratch, LG_G int32_t* err_flag, int32_t bits)
{
    __hip_atomic_fetch_or(hop_scratch + HS_ERROR, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (err_flag) __hip_atomic_fetch_or(err_flag, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ------------------------------------------------------------------------------------------
// Compact position state (legion_core.h): one claim = an ordered-linear-probing insert of
// [epoch | vertex | pending | value] with one atomicMin(u64) per probe.  `low` is pending | slot for a
// sampled neighbour, the final position for a seed.  Whoever is merged away (same vertex, larger value)
// is a slot that lost its first touch: it gets the hop's mark and, in slot_pos, the final position or
// -2 - (the slot it lost to) -- written by the one thread that saw the merge.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void table_claim(LG_G unsigned long long* tab, uint32_t mask, const PosFmt& pf, int32_t id,
                                            uint32_t low, LG_G int32_t* slot_mark, LG_G int32_t* slot_pos, int32_t mark_tag,
                                            LG_G int32_t* hop_scratch, LG_G int32_t* err_flag)
{
    unsigned long long w = lg_tab_word(pf, id, low);
    uint32_t p = lg_tab_hash(id) & mask;
    const int sh = pf.vb + 1;
    const uint32_t lowmask = pf.pending | pf.vmask;
    for (uint32_t it = 0; it <= mask; it++) {
        const unsigned long long old = __hip_atomic_fetch_min(tab + p, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!lg_tab_current(pf, old)) return;                 // took a stale or empty word's place
        if (((old ^ w) >> sh) == 0) {                         // the same vertex: the lower value stays
            const uint32_t a = (uint32_t)old & lowmask, b = (uint32_t)w & lowmask;
            const uint32_t surv = a < b ? a : b, elim = a < b ? b : a;
            if ((elim & pf.pending) && slot_mark != nullptr) {
                const int32_t loser = (int32_t)(elim & pf.vmask);
                slot_mark[loser] = mark_tag;
                slot_pos[loser] = (surv & pf.pending) ? -2 - (int32_t)(surv & pf.vmask) : (int32_t)(surv & pf.vmask);
            }
            return;
        }
        if (old > w) w = old;                                 // displaced a larger word: carry it on
        p = (p + 1) & mask;
    }
    raise_error(hop_scratch, err_flag, LG_ERR_TABLE_FULL);
}

// where the (present) vertex lives: plain loads, the table is not being claimed while this runs
__device__ __forceinline__ uint32_t table_find(const LG_G unsigned long long* tab, uint32_t mask, const PosFmt& pf, int32_t id)
{
    uint32_t p = lg_tab_hash(id) & mask;
    const unsigned long long want = lg_tab_word(pf, id, 0) >> (pf.vb + 1);
    for (uint32_t it = 0; it <= mask; it++) {
        if ((tab[p] >> (pf.vb + 1)) == want) return p;
        p = (p + 1) & mask;
    }
    return 0xFFFFFFFFu;
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
            const PosFmt pf = lg_pos_fmt(L.hop_scratch[HS_EPOCH], L.hop_scratch[HS_VALUE_BITS]);
            if (L.pos_table != nullptr)
                table_claim(L.pos_table, L.pos_mask, pf, src_id, (uint32_t)idx, nullptr, nullptr, 0, L.hop_scratch, L.err_flag);
            else if (L.position_map != nullptr)       // (lds form: no per-vertex state, the seeds are re-read from sampled_ids)
                __hip_atomic_fetch_min(L.position_map + src_id, pf.hi | (uint32_t)idx,
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // seeds are unique (":26 assume no duplicate")
            L.labels[idx] = p.all_labels[at % p.total_cap];
        }
    }
}

void launch_batch_generate(hipStream_t s, const SeedParams& p, const LanePtrs* d_lanes, int32_t n_lanes)
{
    const int32_t n = p.batch_size > 16 ? p.batch_size : 16;
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
    bool last_hop, is_presc, loser_in_dst, compact_hoist;
    LG_G unsigned long long* edge_access_time;
    LG_G unsigned long long* topo_transactions;
    // the lane's buffers, in the global address space (see LG_G in legion_core.h)
    LG_G int32_t* sampled_ids; LG_G int32_t* agg_src_ids; LG_G int32_t* agg_dst_ids; LG_G int32_t* agg_src_off; LG_G int32_t* agg_dst_off;
    LG_G char* tmp_part_ind; LG_G uint32_t* position_map; LG_G int32_t* node_counter; LG_G int32_t* edge_counter;
    LG_G int32_t* slot_dst; LG_G int32_t* slot_pos; LG_G int32_t* slot_mark; LG_G int32_t* slot_fs; LG_G int32_t* node_slot; LG_G unsigned long long* tile_state; LG_G int32_t* hop_scratch;
    LG_G RowHdr* fh_edge;
    LG_G unsigned long long* pos_table; uint32_t pos_mask; LG_G int32_t* err_flag;
    LG_G unsigned long long* claim_pairs; LG_G int32_t* run_off; LG_G int32_t* claim_cnt; int32_t claim_cap, ids_cap;
    LG_G unsigned long long* known_pairs; LG_G int32_t* known_cnt; int32_t known_cap;
    PosFmt pf;
    int32_t mark_tag;   // (epoch, hop): what slot_mark holds for a slot that lost its first touch in THIS hop
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
    a.col_full = LG_GPTR(const int32_t, p.col_full); a.colx_full = LG_GPTR(const lg_v2i, p.colx_full); a.row_hdr = LG_GPTR(const RowHdr, p.row_hdr); a.last_hop = p.last_hop; a.is_presc = p.is_presc; a.loser_in_dst = p.loser_in_dst; a.compact_hoist = p.compact_hoist;
    a.edge_access_time = LG_GPTR(unsigned long long, p.edge_access_time);
    a.topo_transactions = LG_GPTR(unsigned long long, p.topo_transactions);
    a.sampled_ids = LG_GPTR(int32_t, L.sampled_ids); a.agg_src_ids = LG_GPTR(int32_t, L.agg_src_ids);
    a.agg_dst_ids = LG_GPTR(int32_t, L.agg_dst_ids); a.agg_src_off = LG_GPTR(int32_t, L.agg_src_off);
    a.agg_dst_off = LG_GPTR(int32_t, L.agg_dst_off); a.tmp_part_ind = LG_GPTR(char, L.tmp_part_ind);
    a.position_map = LG_GPTR(uint32_t, L.position_map); a.node_counter = LG_GPTR(int32_t, L.node_counter);
    a.edge_counter = LG_GPTR(int32_t, L.edge_counter); a.slot_dst = LG_GPTR(int32_t, L.slot_dst);
    a.slot_pos = LG_GPTR(int32_t, L.slot_pos); a.slot_mark = LG_GPTR(int32_t, L.slot_mark);
    a.slot_fs = LG_GPTR(int32_t, L.slot_fs); a.node_slot = LG_GPTR(int32_t, L.node_slot); a.tile_state = LG_GPTR(unsigned long long, L.tile_state); a.hop_scratch = LG_GPTR(int32_t, L.hop_scratch);
    a.fh_edge = LG_GPTR(RowHdr, L.fh_edge);
    a.pos_table = LG_GPTR(unsigned long long, L.pos_table); a.pos_mask = L.pos_table_mask;
    a.err_flag = LG_GPTR(int32_t, L.err_flag);
    a.claim_pairs = LG_GPTR(unsigned long long, L.claim_pairs);
    a.run_off = LG_GPTR(int32_t, L.run_off);
    a.claim_cnt = LG_GPTR(int32_t, L.claim_cnt); a.claim_cap = L.claim_cap; a.ids_cap = L.ids_cap;
    a.known_pairs = LG_GPTR(unsigned long long, L.known_pairs); a.known_cnt = LG_GPTR(int32_t, L.known_cnt); a.known_cap = L.known_cap;
    a.pf = lg_pos_fmt(a.hop_scratch[HS_EPOCH], a.hop_scratch[HS_VALUE_BITS]);
    a.mark_tag = (a.hop_scratch[HS_EPOCH] << 8) | (p.op_id / INTRABATCH_CON);
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
// LATER (with !SINGLE): the kernel stops after its first sweep -- neighbours in slot_dst, the partition tile's bucket offsets in
// run_off -- and place_kernel writes the pairs (see there)
template <int FORM, int BB, bool SINGLE, bool LATER = false>      // 0 direct array, 1 table, 2 lds with 2^BB buckets per lane; SINGLE: partition tile = super tile
__global__ __launch_bounds__(LG_TILE) __attribute__((amdgpu_num_sgpr(LG_SAMPLE_SGPRS))) void sample_kernel(HopParams hp, const LanePtrs* __restrict__ lanes)
{
    constexpr bool TABLE = FORM == 1;
    constexpr int NB = 1 << BB;
    const int32_t K = SINGLE ? 1 : hp.lds_k;      // super tiles per partition tile
    static_assert(NB <= LG_TILE, "one thread per bucket in the prefix");
    const SampleArgs a = lane_args(hp, lanes);
    __shared__ RowHdr s_hdr[LG_SUPER];
    __shared__ int32_t s_bcnt[NB], s_boff[NB], s_base, s_wtot[LG_TILE / 64];

    const HopGeom g = hop_geometry(a);
    const int32_t tid = threadIdx.x;
    const int32_t count = a.count;
    const bool seeds = (a.op_id == INTRABATCH_CON);
    const LG_G RowHdr* fh = a.fh_edge + g.frontier_off;

    // lds form: the claims of K consecutive super tiles (a partition tile) go, grouped by bucket, into ONE run of the lane's
    // pair array.  SINGLE (K = 1): ranks are taken while the super tile is sampled.  Otherwise (many buckets: a run must stay long
    // enough per bucket to be read in whole sectors): the first sweep samples and counts, the second re-reads slot_dst
    // (this workgroup's own stores) and places the pairs.
    const int32_t nparts = (g.nsuper + K - 1) / K;
    for (int32_t m = blockIdx.x; m < nparts; m += gridDim.x) {
        if (FORM == 2 && !SINGLE && tid < NB) s_bcnt[tid] = 0;      // (visible after the first barrier of the first super tile)
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

            if (FORM == 2 && SINGLE && tid < NB) s_bcnt[tid] = 0;     // (made visible by the barrier above the loads' use below)
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
                            if (a.colx_full != nullptr) {
                                const lg_v2i e = a.colx_full[at];
                                dst[u] = e.x;
                                fs[u] = e.y;
                            } else {
                                dst[u] = a.col_full[at];                        // :239-243
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
                    if (FORM == 2) {
                        // lds form: no claim here; the pair goes to its hash bucket below
                        if (dst[u] >= 0 && a.edge_access_time)
                            __hip_atomic_fetch_add(a.edge_access_time + g.frontier[idx / count], 1ull, __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_AGENT);
                        if (dst[u] < 0) dst[u] = -1;
                    } else if (dst[u] >= 0) {                                  // :244
                        // First touch goes to the LOWEST slot that sampled the vertex.  The atomic returns what
                        // it replaced, so every loser is known without a second look at the state array:
                        //   old < key : the vertex is already in the batch (final position) or a lower slot of this
                        //               hop holds it -> this slot lost, and knows to whom;
                        //   old > key, this epoch : old is a higher slot that held it until now -> THAT slot lost to
                        //               this one (it wrote nothing itself, so the two stores below have one writer);
                        //   otherwise : untouched so far; this slot holds it unless a lower one shows up.
                        // A loser gets the hop's tag in slot_mark and, in slot_pos, the final position or -2 - (the
                        // slot it lost to); the chain of losers ends at the winner (localise follows it).
                        if (TABLE) {      // compact form: the same outcome through the lane's open-addressing table
                            table_claim(a.pos_table, a.pos_mask, a.pf, dst[u], a.pf.pending | (uint32_t)idx, a.slot_mark, a.slot_pos,
                                        a.mark_tag, a.hop_scratch, a.err_flag);
                        } else {
                            const uint32_t key = a.pf.hi | a.pf.pending | (uint32_t)idx;
                            const uint32_t old = __hip_atomic_fetch_min(a.position_map + dst[u], key, __ATOMIC_RELAXED,
                                                                        __HIP_MEMORY_SCOPE_AGENT);
                            if (old < key) {
                                a.slot_mark[idx] = a.mark_tag;
                                a.slot_pos[idx] = (old & a.pf.pending) ? -2 - (int32_t)(old & a.pf.vmask) : (int32_t)(old & a.pf.vmask);
                            } else if ((old & ~(a.pf.pending | a.pf.vmask)) == a.pf.hi) {
                                const int32_t loser = (int32_t)(old & a.pf.vmask);
                                a.slot_mark[loser] = a.mark_tag;
                                a.slot_pos[loser] = -2 - idx;
                            }
                        }
                        if (a.edge_access_time)                                // :358
                            __hip_atomic_fetch_add(a.edge_access_time + g.frontier[idx / count], 1ull, __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_AGENT);
                    } else {
                        dst[u] = -1;
                    }
                    // slot_dst / slot_fs are read once, by the compaction two kernels later: non-temporal stores (+0.5 % on the whole job,
                    // tools/lds_tuning/value_rounds.sh: they do not push the column arrays' lines out of the caches)
                    __builtin_nontemporal_store(dst[u], &a.slot_dst[idx]);
                    if (FORM == 2) a.slot_pos[idx] = -1;      // "no position yet": compact_kernel publishes a first touch's position here (plain store: no difference)
                    if (a.slot_fs != nullptr && dst[u] >= 0) __builtin_nontemporal_store(fs[u], &a.slot_fs[idx]);     // (read for first-touch slots only)
                }
            }
            if (FORM == 2 && SINGLE) {
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

            if (FORM == 2 && !SINGLE) {
#pragma unroll
                for (int u = 0; u < LG_SLOTS_PER_LANE; u++)
                    if (dst[u] >= 0) atomicAdd(&s_bcnt[lg_tab_hash(dst[u]) & (NB - 1)], 1);
            }
            __syncthreads();
        }
        if (FORM == 2 && !SINGLE) {
            // exclusive prefix of the bucket counts over the workgroup, one global reservation, then the second sweep
            const int32_t c = tid < NB ? s_bcnt[tid] : 0;
            int32_t inc = c;
            for (int d = 1; d < 64; d <<= 1) { const int32_t o = __shfl_up(inc, d); if ((tid & 63) >= d) inc += o; }
            if ((tid & 63) == 63) s_wtot[tid >> 6] = inc;
            __syncthreads();
            int32_t wbase = 0, tot = 0;
            for (int w = 0; w < LG_TILE / 64; w++) { if (w < (tid >> 6)) wbase += s_wtot[w]; tot += s_wtot[w]; }
            if (tid < NB) { s_boff[tid] = wbase + inc - c; s_bcnt[tid] = 0; }
            if (tid == 0) {
                s_base = tot > 0 ? __hip_atomic_fetch_add(a.hop_scratch + HS_PAIR_CURSOR, tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
                a.run_off[(int64_t)m * (NB + 1) + NB] = s_base + tot;
            }
            __syncthreads();
            if (tid < NB) a.run_off[(int64_t)m * (NB + 1) + tid] = s_base + s_boff[tid];
            for (int32_t sub = 0; sub < (LATER ? 0 : K); sub++) {
                const int32_t idx0 = (m * K + sub) * LG_SUPER;
                if (idx0 >= g.total) break;
                int32_t d[LG_SLOTS_PER_LANE];
#pragma unroll
                for (int u = 0; u < LG_SLOTS_PER_LANE; u++) {
                    const int32_t idx = idx0 + u * LG_TILE + tid;
                    d[u] = idx < g.total ? a.slot_dst[idx] : -1;
                }
#pragma unroll
                for (int u = 0; u < LG_SLOTS_PER_LANE; u++) {
                    if (d[u] < 0) continue;
                    const int32_t bk = (int32_t)(lg_tab_hash(d[u]) & (NB - 1));
                    const int32_t r = atomicAdd(&s_bcnt[bk], 1);
                    a.claim_pairs[s_base + s_boff[bk] + r] =
                        ((unsigned long long)(uint32_t)d[u] << 32) | (uint32_t)(idx0 + u * LG_TILE + tid);
                }
            }
            __syncthreads();                       // the next partition tile zeroes s_bcnt
        }
    }
}

// ------------------------------------------------------------------------------------------
// K1a (lds form, 64- and 256-bucket classes, partition tiles of at most LG_PLACE_MAX_K super tiles): the second sweep of the
// sampling kernel as a kernel of its own.  With 64-256 buckets a wave's 64 pairs go to ~50 different runs: written straight
// to memory they are 8-byte stores all over the partition tile's region, and the sweep cost the hop-3 launch of [15,10,5] at
// B = 8000 as much again as its scattered column loads.  But a partition tile's pairs, grouped by bucket, form ONE
// contiguous block of the lane's pair array: here they are ranked and staged in LDS (8 KB per super tile) and the block is
// then streamed out, fully coalesced.  The sampling kernel keeps its occupancy for the scattered loads (no staging there);
// this kernel reads slot_dst and writes the pairs as plain streams.
// ------------------------------------------------------------------------------------------
#define LG_PLACE_MAX_K 8
template <int BB>
__global__ __launch_bounds__(LG_TILE) void place_kernel(HopParams hp, const LanePtrs* __restrict__ lanes)
{
    constexpr int NB = 1 << BB;
    extern __shared__ unsigned long long s_stage[];           // hp.lds_k * LG_SUPER pairs
    __shared__ int32_t s_off[NB + 1], s_cnt[NB];
    const SampleArgs a = lane_args(hp, lanes);
    const int32_t K = hp.lds_k;
    const HopGeom g = hop_geometry(a);
    const int32_t tid = threadIdx.x;
    const int32_t nparts = (g.nsuper + K - 1) / K;
    for (int32_t m = blockIdx.x; m < nparts; m += gridDim.x) {
        for (int32_t i = tid; i <= NB; i += LG_TILE) s_off[i] = a.run_off[(int64_t)m * (NB + 1) + i];
        for (int32_t i = tid; i < NB; i += LG_TILE) s_cnt[i] = 0;
        __syncthreads();
        const int32_t base = s_off[0], tot = s_off[NB] - base;
        for (int32_t sub = 0; sub < K; sub++) {
            const int32_t idx0 = (m * K + sub) * LG_SUPER;
            if (idx0 >= g.total) break;
            int32_t d[LG_SLOTS_PER_LANE];
#pragma unroll
            for (int u = 0; u < LG_SLOTS_PER_LANE; u++) {
                const int32_t idx = idx0 + u * LG_TILE + tid;
                d[u] = idx < g.total ? a.slot_dst[idx] : -1;
            }
#pragma unroll
            for (int u = 0; u < LG_SLOTS_PER_LANE; u++) {
                if (d[u] < 0) continue;
                const int32_t bk = (int32_t)(lg_tab_hash(d[u]) & (NB - 1));
                const int32_t r = atomicAdd(&s_cnt[bk], 1);
                s_stage[s_off[bk] - base + r] = ((unsigned long long)(uint32_t)d[u] << 32) | (uint32_t)(idx0 + u * LG_TILE + tid);
            }
        }
        __syncthreads();
        for (int32_t i = tid; i < tot; i += LG_TILE) a.claim_pairs[base + i] = s_stage[i];
        __syncthreads();                           // (the next partition tile re-uses the stage and the counts)
    }
}

// ------------------------------------------------------------------------------------------
// K1b (lds form): de-duplication of a hop's claims, one workgroup per (bucket, lane), entirely in LDS.
//   table word = [ vertex : 32 | pending : 1 | value : 31 ], empty = all ones, ordered linear probing with atomicMin.
//   1. the batch's known vertices that hash into this bucket go in with their position: the seeds from sampled_ids, the
//      nodes earlier hops added from the bucket's list (list_known_kernel) -- or all of them from sampled_ids when the
//      list outgrew its capacity;
//   2. the bucket's claims go in as pending | slot: per vertex the lowest value survives -- a known position beats any
//      slot, a lower slot beats a higher one;
//   3. every claim looks its vertex up: the claim that IS the surviving word is a first touch and stays unmarked; every
//      other claim gets the hop's mark and, in slot_pos, the final position or -2 - (the winning slot) -- exactly what
//      the atomics of the other two forms leave (here the chain of losers always has length one).
// A bucket whose vertices cannot fit the table is processed in P passes over sub-buckets (further hash bits), so the
// result never depends on how the hash spreads the batch.  Nothing survives the hop: nothing to clear, no state that
// scales with the graph.  The claims arrive as ONE LIST PER BUCKET in the 8- and 16-bucket classes (round 4: the workgroup's
// reads then all leave in one round trip, see LISTS below), and as one segment per partition tile of the sampling kernel
// (run_off) in the 64- and 256-bucket classes, where a workgroup addresses claim k of its bucket through the prefix of the
// segment lengths.
// ------------------------------------------------------------------------------------------
#define LG_DEDUP_BATCH 4             // known vertices a thread loads before it works on them (their loads are in flight together)
#ifndef LG_DEDUP_CLAIMS
#define LG_DEDUP_CLAIMS 5           // claims a thread keeps in registers (a bucket of at most LG_DEDUP_CLAIMS * LG_DEDUP_THREADS is "resident")
#endif
#ifndef LG_DEDUP_THREADS
#define LG_DEDUP_THREADS 1024
#endif
__device__ __forceinline__ uint32_t lds_slot_of(uint32_t h) { return (h * 0x9E3779B1u) >> (32 - LG_LDS_TABLE_BITS); }

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

// ---- 8- and 16-bucket classes: one claim list per bucket -----------------------------------------------------------------
// A workgroup's life used to be a chain of dependent round trips to memory (lane pointers -> live counters -> segment table ->
// claims); with one list per bucket, what a thread reads first -- its claims of the list, its seed, its entry of the known list --
// sits at addresses that do not depend on the live counters, so these loads leave TOGETHER with the loads of the counters and
// list lengths (entries past the live lengths are stale and are masked when they are used).  A workgroup can take UNITS buckets
// of its lane in turn, the loads of the next bucket requested before the table work of the current one (the barriers order LDS
// only -- lds_barrier -- so they do not wait for those loads); the default is ONE bucket per workgroup:
// Measured on the 512-lane group of the headline workload (tools/lds_tuning/dedup_ab.sh, timing-only builds): launching 4096
// workgroups of 16 waves costs 54 us before any of them does anything, with all reads requested up front and one barrier it is
// 108 us, the table work brings it to ~190.  Fewer, longer-lived workgroups remove launch cost and hide the loads -- and lose
// under the weave, where the heavy stream's kernel gets its share of the machine by asking for slots again and again while
// the low-priority stream's workgroups take every slot a long-lived workgroup cannot ask for again: two buckets per workgroup
// 209 us against 192; a persistent launch of 512 workgroups over all units (fixed stride or by ticket) 161 us ALONE but
// 240-305 us under the weave.  Also rejected: a table whose words never move (compare-and-swap + min, one LDS load per
// look-up): +25 us; 512-thread workgroups (8 or 16 buckets): +0..40 us.
#ifndef LG_DEDUP_UNITS
#define LG_DEDUP_UNITS 1
#endif
#ifndef LG_DEDUP_ONE_WG_SLOTS
#define LG_DEDUP_ONE_WG_SLOTS 32768    // hops of at most this many slots (in groups of at least LegionTuning.lds_one_wg_lanes lanes): one workgroup per lane
#endif
template <int BB, int UNITS, int CL>      // CL: claims a thread keeps in registers (a bucket of at most CL * LG_DEDUP_THREADS claims is "resident")
__global__ __launch_bounds__(LG_DEDUP_THREADS) __attribute__((amdgpu_num_sgpr(80)))
void dedup_lists_kernel(HopParams hp, const LanePtrs* __restrict__ lanes)
{
    constexpr int NB = 1 << BB;
    constexpr int STEP = NB / UNITS;
    constexpr uint32_t PENDING = 0x80000000u;
    static_assert(NB % UNITS == 0, "buckets per workgroup");
    __shared__ unsigned long long s_tab[LG_LDS_TABLE];
    __shared__ int32_t s_full;
    const int32_t tid = threadIdx.x;
    const SampleArgs a = lane_args(hp, lanes);

    struct Req {                                   // what a unit (bucket) reads first
        unsigned long long rp[CL];    // the thread's claims u * THREADS + tid of the bucket's list
        unsigned long long kl0;                    // the bucket's known list [tid]
        int32_t n_listed, n_claims;
    };
    auto request = [&](int32_t b, Req& q) {
#pragma unroll
        for (int u = 0; u < CL; u++) {
            const int32_t k = u * LG_DEDUP_THREADS + tid;
            q.rp[u] = k < a.claim_cap ? a.claim_pairs[lg_claim_at<NB>(b, k)] : ~0ull;
        }
        q.kl0 = (a.known_pairs != nullptr && tid < a.known_cap) ? a.known_pairs[(int64_t)b * a.known_cap + tid] : ~0ull;
        q.n_listed = a.known_pairs != nullptr ? a.known_cnt[b] : 0;
        q.n_claims = a.claim_cnt[b * LG_CLAIM_CNT_STRIDE];
    };
    auto insert = [&](unsigned long long w, uint32_t h) {
        uint32_t p = lds_slot_of(h);
        for (int it = 0; it < LG_LDS_TABLE; it++) {
            const unsigned long long old = __hip_atomic_fetch_min(&s_tab[p], w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (old == ~0ull || (uint32_t)(old >> 32) == (uint32_t)(w >> 32)) return;    // placed, or merged with the same vertex
            if (old > w) w = old;                                                          // displaced a larger word: carry it on
            p = (p + 1) & (LG_LDS_TABLE - 1);
        }
        s_full = 1;
    };

    Req nxt;
    request((int32_t)blockIdx.x, nxt);
    // the lane's part (the same for every bucket): the first seed of the thread and the node counts before this hop
    const int32_t kid0 = tid < a.ids_cap ? a.sampled_ids[tid] : -1;
    const int32_t n_known = a.node_counter[0] + a.node_counter[1];
    const int32_t n_seed = min(max(a.node_counter[INTRABATCH_CON * 3], 0), n_known);

#pragma unroll 1
    for (int j = 0; j < UNITS; j++) {
        const int32_t b = (int32_t)blockIdx.x + j * STEP;
        Req cur = nxt;
        if (j + 1 < UNITS) request(b + STEP, nxt);

        // the batch's vertices before this hop are the seeds (sampled_ids) and the nodes earlier hops added (the bucket's known
        // list -- or sampled_ids too when there is no list or it outgrew its capacity)
        const int32_t n_listed = cur.n_listed;
        const bool listed = a.known_pairs != nullptr && n_listed <= a.known_cap;
        const int32_t n_scan = listed ? n_seed : n_known;
        const int32_t total = cur.n_claims;            // (the count of the bucket's claims even when the list could not take them all)
        const LG_G unsigned long long* klist = a.known_pairs + (int64_t)b * a.known_cap;
#pragma unroll
        for (int u = 0; u < CL; u++)
            if (u * LG_DEDUP_THREADS + tid >= total) cur.rp[u] = ~0ull;
        if (!listed || tid >= n_listed) cur.kl0 = ~0ull;
        // passes over sub-buckets: see dedup_lds_kernel
        const int32_t known_est = (listed ? n_listed : 0) + n_scan / NB + n_scan / (4 * NB) + 32;
        int32_t passes = 1;
        while ((int64_t)known_est + total > (int64_t)passes * (LG_LDS_TABLE / 16 * LG_LDS_FILL_16THS)) passes <<= 1;
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
                    if (d < 0) continue;
                    if (a.loser_in_dst) d &= ~LG_LOSER_BIT;          // (an earlier pass may have marked the slot)
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
                for (int32_t i = tid; i < LG_LDS_TABLE; i += LG_DEDUP_THREADS) s_tab[i] = ~0ull;
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
                        const unsigned long long pr = i0 == 0 ? cur.kl0 : (i < n_listed ? klist[i] : ~0ull);
                        if (pr == ~0ull) continue;
                        const uint32_t h = lg_tab_hash((int32_t)(pr >> 32));
                        if (((h >> BB) & pmask) != pass) continue;
                        insert(pr, h);
                    }
                for (int32_t k0 = 0; k0 < n_src; k0 += CL * LG_DEDUP_THREADS) {
                    unsigned long long pr[CL];
                    if (resident) {
#pragma unroll
                        for (int u = 0; u < CL; u++) pr[u] = cur.rp[u];
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
#ifndef LG_LDS_NO_RETRY
                if (s_full != 0) { overflow = true; break; }       // (uniform: read behind the barrier, reset behind the next one)
#endif
                for (int32_t k0 = 0; k0 < n_src; k0 += CL * LG_DEDUP_THREADS) {
                    unsigned long long pr[CL];
                    if (resident) {
#pragma unroll
                        for (int u = 0; u < CL; u++) pr[u] = cur.rp[u];
                    } else
                        fetch(k0, pr);
#pragma unroll
                    for (int u = 0; u < CL; u++) {
                        if (pr[u] == ~0ull) continue;
                        const uint32_t id = (uint32_t)(pr[u] >> 32), slot = (uint32_t)pr[u];
                        const uint32_t h = lg_tab_hash((int32_t)id);
                        if (((h >> BB) & pmask) != pass) continue;
                        uint32_t p = lds_slot_of(h);
                        uint32_t v = 0xFFFFFFFFu;
                        for (int it = 0; it < LG_LDS_TABLE; it++) {
                            const unsigned long long w = s_tab[p];
                            if ((uint32_t)(w >> 32) == id) { v = (uint32_t)w; break; }
                            p = (p + 1) & (LG_LDS_TABLE - 1);
                        }
                        if (v != (PENDING | slot)) {          // not the lowest slot of a new vertex
                            if (a.loser_in_dst) a.slot_dst[slot] = (int32_t)(id | LG_LOSER_BIT);      // (vertex ids < 2^30: the mark rides in the id)
                            else a.slot_mark[slot] = a.mark_tag;
                            a.slot_pos[slot] = (v & PENDING) ? -2 - (int32_t)(v & ~PENDING) : (int32_t)v;
                        }
                    }
                }
                lds_barrier();                                 // (the next pass / bucket clears the table)
            }
            if (!overflow) break;
            if (passes >= (1 << 14)) {                             // 2^14 sub-buckets of one bucket still too full: not a hash problem
                if (tid == 0) raise_error(a.hop_scratch, a.err_flag, LG_ERR_TABLE_FULL);
                break;
            }
            passes <<= 1;
        }
        if (tid == 0) a.claim_cnt[b * LG_CLAIM_CNT_STRIDE] = 0;      // the next hop's sampling starts an empty list
    }
}

// ---- 64- and 256-bucket classes: the claims of a bucket arrive as one segment per partition tile of the sampling kernel
//      (run_off); a workgroup addresses claim k of its bucket through the prefix of the segment lengths ------------------------
template <int BB>
__global__ __launch_bounds__(LG_DEDUP_THREADS) __attribute__((amdgpu_num_sgpr(80)))
void dedup_lds_kernel(HopParams hp, const LanePtrs* __restrict__ lanes)
{
    constexpr int NB = 1 << BB;
    const int32_t K = hp.lds_k;                           // super tiles per partition tile
    constexpr int MAX_PARTS = LG_LDS_MAX_PARTS + 2;       // partition tiles of a hop (sample_kernel's K super tiles each)
    const SampleArgs a = lane_args(hp, lanes);
    __shared__ unsigned long long s_tab[LG_LDS_TABLE];
    __shared__ int32_t s_pref[MAX_PARTS];                  // exclusive prefix of the bucket's segment lengths
    __shared__ int32_t s_seg[MAX_PARTS];                   // where the bucket's segment of partition tile t starts in claim_pairs
    __shared__ int32_t s_total, s_full;
    const HopGeom g = hop_geometry(a);
    const int32_t nparts = (g.nsuper + K - 1) / K;
    const int32_t tid = threadIdx.x, b = blockIdx.x;
    const int32_t n_known = a.node_counter[0] + a.node_counter[1];          // nodes of the batch before this hop
    constexpr uint32_t PENDING = 0x80000000u;
    const LG_G int32_t* roff = a.run_off + b;              // roff[t * (NB + 1)]: start of this bucket's segment of partition tile t

    // the batch's vertices before this hop: the seeds are read from sampled_ids, the nodes earlier hops added from the
    // bucket's list (list_known_kernel) -- or from sampled_ids too when there is no list or it outgrew its capacity
    const int32_t n_seed = min(max(a.node_counter[INTRABATCH_CON * 3], 0), n_known);
    const int32_t n_listed = a.known_pairs != nullptr ? a.known_cnt[b] : 0;
    const bool listed = a.known_pairs != nullptr && n_listed <= a.known_cap;
    const int32_t n_scan = listed ? n_seed : n_known;
    const LG_G unsigned long long* klist = a.known_pairs + (int64_t)b * a.known_cap;
    // Nothing below depends on the segment table: the first LG_DEDUP_BATCH known ids / list entries of the thread are loaded
    // now (the usual bucket has no more) and the table is cleared now, while the segment table is being built -- the
    // workgroup's life is a chain of dependent round trips, these two leave it
    int32_t kid[LG_DEDUP_BATCH];
    unsigned long long kl[LG_DEDUP_BATCH];
#pragma unroll
    for (int u = 0; u < LG_DEDUP_BATCH; u++) {
        const int32_t i = u * LG_DEDUP_THREADS + tid;
        kid[u] = i < n_scan ? a.sampled_ids[i] : -1;
        kl[u] = (listed && i < n_listed) ? klist[i] : ~0ull;
    }
    for (int32_t i = tid; i < LG_LDS_TABLE; i += LG_DEDUP_THREADS) s_tab[i] = ~0ull;
    if (tid == 0) s_full = 0;
    bool cleared = true;

    // the bucket's segments, one per partition tile: exclusive prefix of their lengths
    if (tid == 0) s_total = 0;
    for (int32_t t = tid; t < nparts; t += LG_DEDUP_THREADS) {
        const int32_t off = roff[(int64_t)t * (NB + 1)];
        s_pref[t + 1] = roff[(int64_t)t * (NB + 1) + 1] - off;
        s_seg[t] = off;
    }
    __syncthreads();
    if (tid < 64) {                                    // <= 513 entries: wave 0 scans them, a few consecutive entries per lane
        const int32_t per = (nparts + 63) / 64;
        const int32_t lo = min(tid * per, nparts), hi = min(lo + per, nparts);
        int32_t sum = 0;
        for (int32_t t = lo; t < hi; t++) sum += s_pref[t + 1];
        int32_t inc = sum;
        for (int d = 1; d < 64; d <<= 1) { const int32_t o = __shfl_up(inc, d); if (tid >= d) inc += o; }
        int32_t acc = inc - sum;
        for (int32_t t = lo; t < hi; t++) { acc += s_pref[t + 1]; s_pref[t + 1] = acc; }
        if (tid == 0) s_pref[0] = 0;
        if (tid == 63) s_total = inc;
    }
    __syncthreads();
    const int32_t total = s_total;
    // passes: distinct vertices <= known + claims; keep the expected load of a pass at or below LG_LDS_FILL_16THS / 16 of the
    // table.  The bucket's share of the scanned ids is ESTIMATED (an even spread + a quarter; counting it would cost every
    // workgroup one more round trip to memory), and the hash is assumed to spread the bucket evenly over its sub-buckets: when
    // either is wrong (s_full: an insert found no free word) the whole bucket is redone with twice the passes -- every claim's
    // outcome is the same under any partition, so what finished passes already wrote is simply written again.
    const int32_t known_est = (listed ? n_listed : 0) + n_scan / NB + n_scan / (4 * NB) + 32;
    int32_t passes = 1;
    while ((int64_t)known_est + total > (int64_t)passes * (LG_LDS_TABLE / 16 * LG_LDS_FILL_16THS)) passes <<= 1;

    auto segment_of = [&](int32_t k) {                 // claim k of the bucket -> index into claim_pairs
        int32_t lo = 0, hi = nparts;                   // s_pref[lo] <= k < s_pref[hi]
        while (hi - lo > 1) { const int32_t mid = (lo + hi) >> 1; if (s_pref[mid] <= k) lo = mid; else hi = mid; }
        return s_seg[lo] + (k - s_pref[lo]);
    };
    auto insert = [&](unsigned long long w, uint32_t h) {
        uint32_t p = lds_slot_of(h);
        for (int it = 0; it < LG_LDS_TABLE; it++) {
            const unsigned long long old = __hip_atomic_fetch_min(&s_tab[p], w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (old == ~0ull || (uint32_t)(old >> 32) == (uint32_t)(w >> 32)) return;    // placed, or merged with the same vertex
            if (old > w) w = old;                                                          // displaced a larger word: carry it on
            p = (p + 1) & (LG_LDS_TABLE - 1);
        }
        s_full = 1;
    };

    // a bucket of at most LG_DEDUP_BATCH claims per thread (the usual case) keeps them in registers: one trip to memory for both
    // sweeps of every pass
    const bool resident = total <= LG_DEDUP_BATCH * LG_DEDUP_THREADS;
    unsigned long long rp[LG_DEDUP_BATCH];
    auto fetch = [&](int32_t k0, unsigned long long (&pr)[LG_DEDUP_BATCH]) {
#pragma unroll
        for (int u = 0; u < LG_DEDUP_BATCH; u++) {
            const int32_t k = k0 + u * LG_DEDUP_THREADS + tid;
            pr[u] = k < total ? a.claim_pairs[segment_of(k)] : ~0ull;
        }
    };
    if (resident) fetch(0, rp);

  for (;;) {
    const uint32_t pmask = (uint32_t)passes - 1u;
    bool overflow = false;
    for (uint32_t pass = 0; pass <= pmask; pass++) {
        if (!cleared) {
            for (int32_t i = tid; i < LG_LDS_TABLE; i += LG_DEDUP_THREADS) s_tab[i] = ~0ull;
            if (tid == 0) s_full = 0;
            __syncthreads();
        }
        cleared = false;
        for (int32_t i0 = 0; i0 < n_scan; i0 += LG_DEDUP_BATCH * LG_DEDUP_THREADS) {
#pragma unroll
            for (int u = 0; u < LG_DEDUP_BATCH; u++) {
                const int32_t i = i0 + u * LG_DEDUP_THREADS + tid;
                const int32_t id = i0 == 0 ? kid[u] : (i < n_scan ? a.sampled_ids[i] : -1);      // (the first chunk came early)
                if (id < 0) continue;
                const uint32_t h = lg_tab_hash(id);
                if ((h & (NB - 1)) != (uint32_t)b || ((h >> BB) & pmask) != pass) continue;
                insert(((unsigned long long)(uint32_t)id << 32) | (uint32_t)i, h);
            }
        }
        if (listed)
            for (int32_t i0 = 0; i0 < n_listed; i0 += LG_DEDUP_BATCH * LG_DEDUP_THREADS) {
#pragma unroll
                for (int u = 0; u < LG_DEDUP_BATCH; u++) {
                    const int32_t i = i0 + u * LG_DEDUP_THREADS + tid;
                    const unsigned long long pr = i0 == 0 ? kl[u] : (i < n_listed ? klist[i] : ~0ull);
                    if (pr == ~0ull) continue;
                    const uint32_t h = lg_tab_hash((int32_t)(pr >> 32));
                    if (((h >> BB) & pmask) != pass) continue;
                    insert(pr, h);
                }
            }
        for (int32_t k0 = 0; k0 < total; k0 += LG_DEDUP_BATCH * LG_DEDUP_THREADS) {
            unsigned long long pr[LG_DEDUP_BATCH];
            if (resident) {
#pragma unroll
                for (int u = 0; u < LG_DEDUP_BATCH; u++) pr[u] = rp[u];
            } else
                fetch(k0, pr);
#pragma unroll
            for (int u = 0; u < LG_DEDUP_BATCH; u++) {
                if (pr[u] == ~0ull) continue;
                const uint32_t h = lg_tab_hash((int32_t)(pr[u] >> 32));
                if (((h >> BB) & pmask) != pass) continue;
                insert((pr[u] & 0xFFFFFFFF00000000ull) | PENDING | (uint32_t)pr[u], h);
            }
        }
        __syncthreads();
#ifndef LG_LDS_NO_RETRY
        if (s_full != 0) { overflow = true; break; }       // (uniform: read behind the barrier, reset behind the next one)
#endif
        for (int32_t k0 = 0; k0 < total; k0 += LG_DEDUP_BATCH * LG_DEDUP_THREADS) {
            unsigned long long pr[LG_DEDUP_BATCH];
            if (resident) {
#pragma unroll
                for (int u = 0; u < LG_DEDUP_BATCH; u++) pr[u] = rp[u];
            } else
                fetch(k0, pr);
#pragma unroll
            for (int u = 0; u < LG_DEDUP_BATCH; u++) {
                if (pr[u] == ~0ull) continue;
                const uint32_t id = (uint32_t)(pr[u] >> 32), slot = (uint32_t)pr[u];
                const uint32_t h = lg_tab_hash((int32_t)id);
                if (((h >> BB) & pmask) != pass) continue;
                uint32_t p = lds_slot_of(h);
                uint32_t v = 0xFFFFFFFFu;
                for (int it = 0; it < LG_LDS_TABLE; it++) {
                    const unsigned long long w = s_tab[p];
                    if ((uint32_t)(w >> 32) == id) { v = (uint32_t)w; break; }
                    p = (p + 1) & (LG_LDS_TABLE - 1);
                }
                if (v != (PENDING | slot)) {          // not the lowest slot of a new vertex
                    if (a.loser_in_dst) a.slot_dst[slot] = (int32_t)(id | LG_LOSER_BIT);      // (vertex ids < 2^30: the mark rides in the id)
                    else a.slot_mark[slot] = a.mark_tag;
                    a.slot_pos[slot] = (v & PENDING) ? -2 - (int32_t)(v & ~PENDING) : (int32_t)v;
                }
            }
        }
        __syncthreads();
    }
    if (!overflow) break;
    if (passes >= (1 << 14)) {                             // 2^14 sub-buckets of one bucket still too full: not a hash problem
        if (tid == 0) raise_error(a.hop_scratch, a.err_flag, LG_ERR_TABLE_FULL);
        break;
    }
    passes <<= 1;
    __syncthreads();
  }
    if (b == 0 && tid == 0) a.hop_scratch[HS_PAIR_CURSOR] = 0;      // the next hop's sampling starts a new pair array
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
    const bool lds = a.claim_pairs != nullptr;                            // the lds form of the first-touch state
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
        int32_t v[LG_SLOTS_PER_LANE], mk[LG_SLOTS_PER_LANE];
        unsigned long long mv[LG_SLOTS_PER_LANE], mf[LG_SLOTS_PER_LANE];
        const bool inl = a.loser_in_dst;                                  // (uniform) the loser mark rides in slot_dst
#pragma unroll
        for (int u = 0; u < LG_SLOTS_PER_LANE; u++) {
            const int32_t idx = idx0 + u * CT + tid;
            v[u] = idx < total ? a.slot_dst[idx] : -1;
            mk[u] = (idx < total && !inl) ? a.slot_mark[idx] : 0;
        }
        // (compact_hoist) what depends on the slot INDEX only -- the vertex the slot sampled for, its position, the carried cache slot --
        // is loaded together with slot_dst, for every slot of the tile: more bytes (invalid slots too), one dependent round trip less
        int32_t src_of[LG_SLOTS_PER_LANE], src_pos[LG_SLOTS_PER_LANE], fsv[LG_SLOTS_PER_LANE];
        const bool hoist = a.compact_hoist;
        if (hoist) {
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
        }
#pragma unroll
        for (int u = 0; u < LG_SLOTS_PER_LANE; u++) {
            const bool valid = v[u] >= 0;
            const bool first = valid && (inl ? (v[u] & LG_LOSER_BIT) == 0 : mk[u] != a.mark_tag);   // nobody marked it a loser in this hop
            if (valid && inl) v[u] &= ~LG_LOSER_BIT;
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
        uint32_t tab_at[LG_SLOTS_PER_LANE];
        RowHdr nh[LG_SLOTS_PER_LANE];
#pragma unroll
        for (int u = 0; u < LG_SLOTS_PER_LANE; u++) {
            const int32_t idx = idx0 + u * CT + tid;
            if (v[u] >= 0) {
                const bool first = (mf[u] >> lane) & 1ull;
                const int32_t dst = v[u];
                if (!hoist) {
                    const int32_t q = idx / a.count;
                    src_of[u] = frontier[q];
                    // position of the node sampled for == construct_graph's position_map[agg_dst_ids[e]]
                    src_pos[u] = seeds ? q : a.agg_src_off[f_off + q];
                    fsv[u] = (first && a.slot_fs != nullptr) ? a.slot_fs[idx] : LG_FS_UNKNOWN;   // the new node's feature-cache slot, if carried
                } else if (!first) {
                    fsv[u] = LG_FS_UNKNOWN;
                }
                if (!LAST) nh[u] = load_hdr(a.row_hdr + dst);      // next hop's frontier header
                lost_pos[u] = first ? 0 : a.slot_pos[idx];               // final already, or -2 - (slot it lost to)
                tab_at[u] = (first && !LAST && a.pos_table != nullptr) ? table_find(a.pos_table, a.pos_mask, a.pf, dst) : 0u;
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
        // lds form: the first touches' positions are known now -- publish them before anything else, later super tiles' losers
        // are waiting for nothing but this (publishing with the other stores below would chain every tile's loads behind the
        // stores of the tiles before it)
        int32_t n_at[LG_SLOTS_PER_LANE];
#pragma unroll
        for (int u = 0; u < LG_SLOTS_PER_LANE; u++) {
            n_at[u] = -1;
            if ((mf[u] >> lane) & 1ull) {
                n_at[u] = node_base + xn + s_pre[1][u * (CT / 64) + wave] + __popcll(mf[u] & lt);
                if (lds) __hip_atomic_store(a.slot_pos + idx0 + u * CT + tid, n_at[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (lds) {
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
            const int32_t idx = idx0 + u * CT + tid;
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
                // :271 -- later hops look the position up in the state array; after the last hop nobody
                // does, and same-hop duplicates resolve through slot_pos (a small, cache-resident array)
                if (!LAST) {
                    if (a.pos_table == nullptr) {
                        if (a.position_map != nullptr) a.position_map[dst] = (int32_t)(a.pf.hi | (uint32_t)n);   // (lds form: none)
                    } else if (tab_at[u] != 0xFFFFFFFFu)
                        a.pos_table[tab_at[u]] = lg_tab_word(a.pf, dst, (uint32_t)n);
                    else
                        raise_error(a.hop_scratch, a.err_flag, LG_ERR_TABLE_FULL);
                }
                if (!lds) a.slot_pos[idx] = n;                     // (atomics forms: what localise follows)
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

// ------------------------------------------------------------------------------------------
// K5: construct_graph's neighbour side
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(LG_TILE) void localise_kernel(HopParams hp, const LanePtrs* __restrict__ lanes)
{
    const SampleArgs a = lane_args(hp, lanes);
    const LG_G int32_t* hs = a.hop_scratch;
    const int32_t n_edge = hs[HS_N_EDGE], edge_base = hs[HS_EDGE_BASE];
    const int32_t nsuper = (n_edge + LG_SUPER - 1) / LG_SUPER;
    // scatter already localised every edge whose neighbour was final or first-touched by that very
    // slot; what is left (< 0) are neighbours owned by ANOTHER slot of this hop, whose new position
    // scatter left in slot_pos[winner]: a short walk through a small array (:289-293)
    for (int32_t st = blockIdx.x; st < nsuper; st += gridDim.x) {
        int32_t cur[LG_SLOTS_PER_LANE];
#pragma unroll
        for (int u = 0; u < LG_SLOTS_PER_LANE; u++) {
            const int32_t e = st * LG_SUPER + u * LG_TILE + threadIdx.x;
            cur[u] = e < n_edge ? a.agg_src_off[edge_base + e] : 0;
        }
#pragma unroll
        for (int u = 0; u < LG_SLOTS_PER_LANE; u++) {
            const int32_t e = st * LG_SUPER + u * LG_TILE + threadIdx.x;
            if (cur[u] < 0) {      // -2 - (slot it lost to); that slot may have lost to a lower one in turn
                int32_t c = cur[u];
                for (int it = 0; it < (1 << 20) && c < -1; it++) c = a.slot_pos[-2 - c];   // strictly descending slots
                if (c < -1) raise_error(a.hop_scratch, a.err_flag, LG_ERR_CHAIN);         // cannot happen: every chain ends at a winner
                a.agg_src_off[edge_base + e] = c;
            }
        }
    }
}

int g_sample_stages = 15;
void launch_random_sample(hipStream_t s, const HopParams& p, const LanePtrs* d_lanes, int32_t n_lanes, int32_t form)
{
    const int stages = g_sample_stages;
    // Fixed grids that stride over super tiles; grid.y = lanes (independent mini-batches of a group).
    int32_t max_super = (p.max_slots + LG_SUPER - 1) / LG_SUPER;
    if (max_super < 1) max_super = 1;
    int32_t gx = max_super < 1024 ? max_super : 1024;
    const int max_wg = tuning().sample_max_wg;
    while (gx > 64 && (int64_t)gx * n_lanes > max_wg) gx /= 2;  // keep the whole launch near 2 x resident capacity
    while (gx > 1 && (int64_t)gx * n_lanes > max_wg && max_wg < 4096) gx /= 2;   // (experiments with fewer workgroups)
    const dim3 grid(gx, n_lanes);
    if (form == 2) {
        // buckets per lane follow the pool's largest hop (legion_core.h); super tiles per partition tile follow THIS hop: at
        // most LG_LDS_MAX_PARTS partition tiles, and no larger than leaves the launch ~8 k workgroups by the hop's capacity
        // (a hop typically fills a quarter of it: ~2 k active ones; measured at B = 8000: 2 k -> 8 k +1...2 %, beyond: the same)
        HopParams q = p;
        const bool small = p.lds_bucket_bits == LG_LDS_BITS_SMALL || p.lds_bucket_bits == LG_LDS_BITS_SMALL16;
        const int32_t k_hi = small ? 1 : (p.lds_bucket_bits == LG_LDS_BITS_MEDIUM ? LG_LDS_K_MEDIUM : LG_LDS_K_LARGE);
        int32_t k_lo = p.lds_bucket_bits == LG_LDS_BITS_LARGE ? 4 : 1;
        while (max_super > LG_LDS_MAX_PARTS * k_lo) k_lo *= 2;
        int32_t k = k_hi > k_lo ? k_hi : k_lo;
        const int want_wg = tuning().lds_part_wg;
        while (k > k_lo && (int64_t)(max_super / k) * n_lanes < want_wg) k /= 2;
        if (!small && k > LG_PLACE_MAX_K && k_lo <= LG_PLACE_MAX_K) k = LG_PLACE_MAX_K;     // (the staged placement takes 8 super tiles at most)
        q.lds_k = k;
        int32_t gp = (max_super + k - 1) / k;                  // one workgroup per partition tile ...
        while (gp > 16 && (int64_t)gp * n_lanes > 16384) gp = (gp + 1) / 2;  // ... within reason
        // a hop of few slots (the first hop of a B = 1024 batch: 25 600, ~400 claims per bucket): its 4096 one-bucket workgroups are all
        // launch (54 us of nothing per 512-lane group); ONE workgroup per lane can take the lane's buckets in turn instead
        // (LegionTuning.lds_one_wg_lanes; off by default: no gain under the weave)
        const int one_wg_lanes = tuning().lds_one_wg_lanes;
        const bool one_wg_per_lane = small && one_wg_lanes > 0 && n_lanes >= one_wg_lanes && p.max_slots <= LG_DEDUP_ONE_WG_SLOTS;
        if (p.lds_bucket_bits == LG_LDS_BITS_SMALL) {
            if (stages & 1) sample_kernel<2, LG_LDS_BITS_SMALL, true><<<grid, LG_TILE, 0, s>>>(q, d_lanes);
            hipCheckError();
            if (stages & 2) {
                if (one_wg_per_lane) dedup_lists_kernel<LG_LDS_BITS_SMALL, 1 << LG_LDS_BITS_SMALL, 2><<<dim3(1, n_lanes), LG_DEDUP_THREADS, 0, s>>>(q, d_lanes);
                else dedup_lists_kernel<LG_LDS_BITS_SMALL, LG_DEDUP_UNITS, LG_DEDUP_CLAIMS><<<dim3((1 << LG_LDS_BITS_SMALL) / LG_DEDUP_UNITS, n_lanes), LG_DEDUP_THREADS, 0, s>>>(q, d_lanes);
            }
        } else if (p.lds_bucket_bits == LG_LDS_BITS_SMALL16) {
            if (stages & 1) sample_kernel<2, LG_LDS_BITS_SMALL16, true><<<grid, LG_TILE, 0, s>>>(q, d_lanes);
            hipCheckError();
            if (stages & 2) {
                if (one_wg_per_lane) dedup_lists_kernel<LG_LDS_BITS_SMALL16, 1 << LG_LDS_BITS_SMALL16, 2><<<dim3(1, n_lanes), LG_DEDUP_THREADS, 0, s>>>(q, d_lanes);
                else dedup_lists_kernel<LG_LDS_BITS_SMALL16, LG_DEDUP_UNITS, LG_DEDUP_CLAIMS><<<dim3((1 << LG_LDS_BITS_SMALL16) / LG_DEDUP_UNITS, n_lanes), LG_DEDUP_THREADS, 0, s>>>(q, d_lanes);
            }
        } else if (p.lds_bucket_bits == LG_LDS_BITS_MEDIUM) {
            if (k <= LG_PLACE_MAX_K) {
                sample_kernel<2, LG_LDS_BITS_MEDIUM, false, true><<<dim3(gp, n_lanes), LG_TILE, 0, s>>>(q, d_lanes);
                hipCheckError();
                place_kernel<LG_LDS_BITS_MEDIUM><<<dim3(gp, n_lanes), LG_TILE, (size_t)k * LG_SUPER * sizeof(unsigned long long), s>>>(q, d_lanes);
            } else
                sample_kernel<2, LG_LDS_BITS_MEDIUM, false><<<dim3(gp, n_lanes), LG_TILE, 0, s>>>(q, d_lanes);
            hipCheckError();
            dedup_lds_kernel<LG_LDS_BITS_MEDIUM><<<dim3(1 << LG_LDS_BITS_MEDIUM, n_lanes), LG_DEDUP_THREADS, 0, s>>>(q, d_lanes);
        } else {
            if (k <= LG_PLACE_MAX_K) {
                sample_kernel<2, LG_LDS_BITS_LARGE, false, true><<<dim3(gp, n_lanes), LG_TILE, 0, s>>>(q, d_lanes);
                hipCheckError();
                place_kernel<LG_LDS_BITS_LARGE><<<dim3(gp, n_lanes), LG_TILE, (size_t)k * LG_SUPER * sizeof(unsigned long long), s>>>(q, d_lanes);
            } else
                sample_kernel<2, LG_LDS_BITS_LARGE, false><<<dim3(gp, n_lanes), LG_TILE, 0, s>>>(q, d_lanes);
            hipCheckError();
            dedup_lds_kernel<LG_LDS_BITS_LARGE><<<dim3(1 << LG_LDS_BITS_LARGE, n_lanes), LG_DEDUP_THREADS, 0, s>>>(q, d_lanes);
        }
    } else if (form == 1) {
        sample_kernel<1, 0, true><<<grid, LG_TILE, 0, s>>>(p, d_lanes);
    } else {
        sample_kernel<0, 0, true><<<grid, LG_TILE, 0, s>>>(p, d_lanes);
    }
    hipCheckError();
    if (!(stages & 4)) return;
    // compaction: LG_COMPACT_THREADS per workgroup (a workgroup iteration takes 4 x that many consecutive slots), as many workgroups per
    // lane as the sampling launch has per 1024 slots' worth
    {
        const dim3 cgrid(std::max(1, (int)grid.x * LG_TILE / LG_COMPACT_THREADS), n_lanes);
        if (p.last_hop) compact_kernel<true, LG_COMPACT_THREADS><<<cgrid, LG_COMPACT_THREADS, 0, s>>>(p, d_lanes);
        else compact_kernel<false, LG_COMPACT_THREADS><<<cgrid, LG_COMPACT_THREADS, 0, s>>>(p, d_lanes);
    }
    hipCheckError();
    if (form == 2 && !p.last_hop) {       // later hops must recognise the nodes this one added: their buckets' lists
        int32_t chunks = (p.max_slots + LG_LIST_CHUNK - 1) / LG_LIST_CHUNK;
        if (chunks > 256) chunks = 256;
        if (p.lds_bucket_bits == LG_LDS_BITS_SMALL) list_known_kernel<LG_LDS_BITS_SMALL><<<dim3(chunks, n_lanes), LG_TILE, 0, s>>>(p, d_lanes);
        else if (p.lds_bucket_bits == LG_LDS_BITS_SMALL16) list_known_kernel<LG_LDS_BITS_SMALL16><<<dim3(chunks, n_lanes), LG_TILE, 0, s>>>(p, d_lanes);
        else if (p.lds_bucket_bits == LG_LDS_BITS_MEDIUM) list_known_kernel<LG_LDS_BITS_MEDIUM><<<dim3(chunks, n_lanes), LG_TILE, 0, s>>>(p, d_lanes);
        else list_known_kernel<LG_LDS_BITS_LARGE><<<dim3(chunks, n_lanes), LG_TILE, 0, s>>>(p, d_lanes);
        hipCheckError();
    }
    if (form != 2) {      // (lds form: scatter placed every loser itself from the first-touch ballots)
        localise_kernel<<<grid, LG_TILE, 0, s>>>(p, d_lanes);
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
// start (:151).  Here nothing is cleared: the lane's epoch goes up by one, which turns every entry
// the batch wrote into "untouched" (see legion_core.h).  Every lg_pos_epoch_max(vb) batches the array
// is refilled with 0xFF by this kernel.  The workgroup that draws the last ticket publishes the
// new epoch (all workgroups have read the old one by then) and advances the device-resident
// iteration used by graph replay.
// ------------------------------------------------------------------------------------------
__global__ void end_of_batch_kernel(const LanePtrs* __restrict__ lanes, int32_t* __restrict__ iter_state)
{
    const BracketLane L = bracket_lane(lanes[blockIdx.y]);
    __shared__ int32_t s_last;
    const int32_t epoch = L.hop_scratch[HS_EPOCH];
    const int32_t epoch_max = lg_pos_epoch_max(L.hop_scratch[HS_VALUE_BITS]);
    if (epoch >= epoch_max) {
        if (L.pos_table != nullptr) {
            for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= (int64_t)L.pos_mask; i += (int64_t)gridDim.x * blockDim.x)
                L.pos_table[i] = ~0ull;
        } else if (L.position_map != nullptr) {
            LG_G uint32_t* pm = L.position_map;
            for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < L.total_num_nodes; i += (int64_t)gridDim.x * blockDim.x)
                pm[i] = 0xFFFFFFFFu;
        }
        // the loser marks carry (epoch, hop): epochs are about to repeat
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < L.max_slots; i += (int64_t)gridDim.x * blockDim.x)
            L.slot_mark[i] = 0;
    }
    // GPURunner's lanes: the batch's counters in host-visible memory, as the trainer end will read them -- node_counter[2..3]
    // already holding what the last gather op leaves there (counter_update(op%3==1), operator_impl.cu:83-85: the range of the
    // last hop's new nodes), whether or not that gather has run yet.  Visible to the host once the launch group has completed.
    if (L.counter_mirror != nullptr && blockIdx.x == 0 && threadIdx.x < 32) {
        const int32_t t = threadIdx.x;
        int32_t v = t < 16 ? L.node_counter[t] : L.edge_counter[t - 16];
        const int32_t hop_num = L.node_counter[INTRABATCH_CON * 3 - 1];
        if (t == 2 && hop_num >= 0 && hop_num <= 6) v = L.hop_scratch[HS_RANGE + 2 * hop_num];
        if (t == 3 && hop_num >= 0 && hop_num <= 6) v = L.hop_scratch[HS_RANGE + 2 * hop_num + 1];
        L.counter_mirror[t] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0)   // no fence needed: the kernel boundary publishes the refill and the new epoch
        s_last = (__hip_atomic_fetch_add(L.hop_scratch + HS_TICKET, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ==
                  (int32_t)gridDim.x - 1) ? 1 : 0;
    __syncthreads();
    if (s_last && threadIdx.x == 0) {
        L.hop_scratch[HS_TICKET] = 0;
        L.hop_scratch[HS_EPOCH] = epoch >= epoch_max ? 1 : epoch + 1;
        if (iter_state != nullptr && blockIdx.y == 0) iter_state[0] += iter_state[1];
    }
}

void launch_end_of_batch(hipStream_t s, const LanePtrs* d_lanes, int32_t n_lanes, int32_t* iter_state,
                         int64_t state_bytes)
{
    // enough workgroups to refill the position state (array or table) at HBM speed on the rare epoch wrap,
    // few enough to cost nothing otherwise
    int32_t gx = (int32_t)((state_bytes + (1 << 20) - 1) >> 20);   // ~1 MiB per workgroup
    if (gx < 1) gx = 1;
    if (gx > 512) gx = 512;
    while (gx > 16 && gx * n_lanes > 2048) gx /= 2;
    end_of_batch_kernel<<<dim3(gx, n_lanes), 256, 0, s>>>(d_lanes, iter_state);
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
