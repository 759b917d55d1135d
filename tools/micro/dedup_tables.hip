// Micro-benchmark behind DESIGN.md 4.2: how fast can a group of mini-batches claim first touches?
//   hipcc --offload-arch=gfx950 -O3 dedup_tables.hip -o dedup_tables && ./dedup_tables
// One "lane" = one mini-batch: KEYS_PER_LANE sampled vertex ids (uniform over N = 2^26, ~12 % repeats),
// LANES lanes per launch (the shape of hop 2 of a 128-lane group at B = 1024: 4.1 M claims).
// Variants, all returning per claim what the sampler needs (who was first, lowest slot wins):
//   A  direct   : atomicMin(u32) on a per-lane uint32[N] array (round-1 design, 268 MB per lane)
//   B  ordered  : per-lane open-addressing table of 64-bit words [id:32 | slot:32], one atomicMin(u64) per probe;
//                 a larger word that loses its place is carried to the next position (ordered linear probing)
//   C  cas      : per-lane table {key u32, val u32}: load key, CAS an empty key, atomicMin(u32) on val
//   D  ordered, workgroup-scope atomics, every lane touched by ONE XCD only (workgroups pick lanes by
//      HW_REG_XCC_ID): tells whether an atomic without sc1 executes in the XCD's L2 and how fast
// Each variant is checked against a CPU answer (per key: lowest slot).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <unordered_map>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__host__ __device__ inline uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

static constexpr int LANES = 128;
static constexpr int KEYS_PER_LANE = 32768;
static constexpr uint32_t N = 1u << 26;

// ---- A ---------------------------------------------------------------------------------------
__global__ void direct_kernel(uint32_t* const* state, const uint32_t* keys, uint32_t* first)
{
    const int lane = blockIdx.y;
    uint32_t* st = state[lane];
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < KEYS_PER_LANE; i += gridDim.x * blockDim.x) {
        const uint32_t k = keys[lane * KEYS_PER_LANE + i];
        const uint32_t old = __hip_atomic_fetch_min(st + k, i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old < i) first[lane * KEYS_PER_LANE + i] = old;                       // lost to a lower slot
        else if (old != 0xFFFFFFFFu) first[lane * KEYS_PER_LANE + old] = i;       // the slot that held it lost to this one
    }
}

// ---- B / D -----------------------------------------------------------------------------------
template <int SCOPE>
__device__ __forceinline__ void ordered_insert(unsigned long long* tab, uint32_t mask, uint32_t key, uint32_t slot,
                                               uint32_t* lost_to, uint32_t* steps)
{
    // lost_to[s] = the lower slot that slot s lost its vertex to (one writer per word: a claim is merged away once)
    unsigned long long w = ((unsigned long long)key << 32) | slot;
    uint32_t p = mix(key) & mask;
    for (uint32_t it = 0; it <= mask; it++) {
        const unsigned long long old = __hip_atomic_fetch_min(tab + p, w, __ATOMIC_RELAXED, SCOPE);
        (*steps)++;
        if (old == ~0ull) break;                                   // empty: placed
        if ((uint32_t)(old >> 32) == (uint32_t)(w >> 32)) {        // same vertex: the lowest slot stays, the other one lost
            const uint32_t a = (uint32_t)old, b = (uint32_t)w;
            lost_to[a > b ? a : b] = a > b ? b : a;
            break;
        }
        if (old > w) w = old;                                      // displaced a larger word: carry it to the next position
        p = (p + 1) & mask;
    }
}

template <int SCOPE, bool XCD_LOCAL>
__global__ void ordered_kernel(unsigned long long* const* tabs, uint32_t mask, const uint32_t* keys, uint32_t* first,
                               uint32_t* lane_ticket, unsigned long long* step_count)
{
    uint32_t steps = 0;
    if (!XCD_LOCAL) {
        const int lane = blockIdx.y;
        for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < KEYS_PER_LANE; i += gridDim.x * blockDim.x)
            ordered_insert<SCOPE>(tabs[lane], mask, keys[lane * KEYS_PER_LANE + i], i, first + lane * KEYS_PER_LANE, &steps);
    } else {
        // lanes l with l % 8 == xcc belong to this XCD; its workgroups share them through a ticket per lane
        const uint32_t xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7u;   // HW_REG_XCC_ID[3:0]
        __shared__ uint32_t s_t;
        for (int lane = xcc; lane < LANES; lane += 8) {
            for (;;) {
                if (threadIdx.x == 0) s_t = atomicAdd(lane_ticket + lane, 1u);
                __syncthreads();
                const uint32_t chunk = s_t;
                __syncthreads();
                if (chunk * blockDim.x >= KEYS_PER_LANE) break;
                const uint32_t i = chunk * blockDim.x + threadIdx.x;
                ordered_insert<SCOPE>(tabs[lane], mask, keys[lane * KEYS_PER_LANE + i], i, first + lane * KEYS_PER_LANE, &steps);
            }
        }
    }
    if (step_count) atomicAdd(step_count, (unsigned long long)steps);
}

// ---- C ---------------------------------------------------------------------------------------
struct KV { uint32_t key, val; };
__global__ void cas_kernel(KV* const* tabs, uint32_t mask, const uint32_t* keys, uint32_t* first)
{
    const int lane = blockIdx.y;
    KV* tab = tabs[lane];
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < KEYS_PER_LANE; i += gridDim.x * blockDim.x) {
        const uint32_t k = keys[lane * KEYS_PER_LANE + i];
        uint32_t p = mix(k) & mask;
        for (uint32_t it = 0; it <= mask; it++) {
            uint32_t cur = __hip_atomic_load(&tab[p].key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (cur == 0xFFFFFFFFu) {
                uint32_t expect = 0xFFFFFFFFu;
                if (__hip_atomic_compare_exchange_strong(&tab[p].key, &expect, k, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                         __HIP_MEMORY_SCOPE_AGENT))
                    cur = k;
                else
                    cur = expect;
            }
            if (cur == k) {
                const uint32_t old = __hip_atomic_fetch_min(&tab[p].val, i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (old < i) first[lane * KEYS_PER_LANE + i] = old;
                else if (old != 0xFFFFFFFFu) first[lane * KEYS_PER_LANE + old] = i;
                break;
            }
            p = (p + 1) & mask;
        }
    }
}

int main()
{
    std::vector<uint32_t> keys((size_t)LANES * KEYS_PER_LANE);
    std::vector<uint32_t> want_min(keys.size());     // per claim: the lowest slot of its vertex in the lane
    for (int l = 0; l < LANES; l++) {
        std::unordered_map<uint32_t, uint32_t> lo;
        for (int i = 0; i < KEYS_PER_LANE; i++) {
            uint32_t r = mix(l * 1000003u + i * 7919u + 12345u);
            uint32_t k = (r % 100 < 12) ? mix(l * 31u + (r >> 8) % 2000u) % N : mix(r + 99u) % N;
            keys[(size_t)l * KEYS_PER_LANE + i] = k;
            auto it = lo.find(k);
            if (it == lo.end()) lo[k] = i;
        }
        for (int i = 0; i < KEYS_PER_LANE; i++) want_min[(size_t)l * KEYS_PER_LANE + i] = lo[keys[(size_t)l * KEYS_PER_LANE + i]];
    }
    uint32_t *d_keys, *d_first, *d_ticket;
    unsigned long long* d_steps;
    CK(hipMalloc(&d_keys, keys.size() * 4));
    CK(hipMalloc(&d_first, keys.size() * 4));
    CK(hipMalloc(&d_ticket, LANES * 4));
    CK(hipMalloc(&d_steps, 8));
    CK(hipMemcpy(d_keys, keys.data(), keys.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    std::vector<uint32_t> got(keys.size());
    const size_t total = keys.size();

    // every claim but the lowest slot of its vertex must end up marked as having lost to a LOWER slot of the same
    // vertex (not necessarily the lowest: the chain of losers ends there); the lowest slot stays unmarked
    auto check = [&](const char* name) {
        CK(hipMemcpy(got.data(), d_first, total * 4, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (size_t j = 0; j < total; j++) {
            const uint32_t i = (uint32_t)(j % KEYS_PER_LANE);
            const bool is_min = want_min[j] == i;
            const bool lost = got[j] != 0xFFFFFFFFu;
            if (is_min == lost) bad++;
            else if (lost && (got[j] >= i || keys[j - i + got[j]] != keys[j])) bad++;   // lost to a lower slot of the same vertex
        }
        printf("    %-10s check: %zu wrong of %zu claims\n", name, bad, total);
    };

    dim3 grid(KEYS_PER_LANE / 256 / 4, LANES);
    // ---- A
    {
        std::vector<uint32_t*> st(LANES);
        for (int l = 0; l < LANES; l++) { CK(hipMalloc(&st[l], (size_t)N * 4)); }
        uint32_t** d_st; CK(hipMalloc(&d_st, LANES * 8));
        CK(hipMemcpy(d_st, st.data(), LANES * 8, hipMemcpyHostToDevice));
        float best = 1e9;
        for (int rep = 0; rep < 3; rep++) {
            for (int l = 0; l < LANES; l++) CK(hipMemsetAsync(st[l], 0xFF, (size_t)N * 4));
            CK(hipMemsetAsync(d_first, 0xFF, total * 4));
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(a));
            direct_kernel<<<grid, 256>>>(d_st, d_keys, d_first);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b)); best = std::min(best, ms);
        }
        printf("A direct  uint32[N] per lane (%d x 268 MB): %7.1f us  %5.1f G claims/s\n", LANES, best * 1e3, total / best / 1e6);
        check("direct");
        for (int l = 0; l < LANES; l++) CK(hipFree(st[l]));
        CK(hipFree(d_st));
    }
    for (int log2t : {16, 17, 19}) {
        const uint32_t T = 1u << log2t, mask = T - 1;
        std::vector<unsigned long long*> tb(LANES);
        for (int l = 0; l < LANES; l++) CK(hipMalloc(&tb[l], (size_t)T * 8));
        unsigned long long** d_tb; CK(hipMalloc(&d_tb, LANES * 8));
        CK(hipMemcpy(d_tb, tb.data(), LANES * 8, hipMemcpyHostToDevice));
        auto reset = [&]() {
            for (int l = 0; l < LANES; l++) CK(hipMemsetAsync(tb[l], 0xFF, (size_t)T * 8));
            CK(hipMemsetAsync(d_ticket, 0, LANES * 4));
            CK(hipMemsetAsync(d_first, 0xFF, total * 4));
            CK(hipMemsetAsync(d_steps, 0, 8));
            CK(hipDeviceSynchronize());
        };
        auto run = [&](int which, const char* name) {
            float best = 1e9;
            unsigned long long steps = 0;
            for (int rep = 0; rep < 3; rep++) {
                reset();
                CK(hipEventRecord(a));
                if (which == 0) ordered_kernel<__HIP_MEMORY_SCOPE_AGENT, false><<<grid, 256>>>(d_tb, mask, d_keys, d_first, d_ticket, d_steps);
                if (which == 1) ordered_kernel<__HIP_MEMORY_SCOPE_AGENT, true><<<1024, 256>>>(d_tb, mask, d_keys, d_first, d_ticket, d_steps);
                if (which == 2) ordered_kernel<__HIP_MEMORY_SCOPE_WORKGROUP, true><<<1024, 256>>>(d_tb, mask, d_keys, d_first, d_ticket, d_steps);
                if (which == 3) cas_kernel<<<grid, 256>>>((KV* const*)d_tb, mask, d_keys, d_first);
                CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
                float ms; CK(hipEventElapsedTime(&ms, a, b)); best = std::min(best, ms);
            }
            CK(hipMemcpy(&steps, d_steps, 8, hipMemcpyDeviceToHost));
            printf("%-44s T=2^%d (%4.1f MB/lane): %7.1f us  %5.1f G claims/s  %.3f atomics/claim\n", name, log2t, T * 8.0 / 1e6,
                   best * 1e3, total / best / 1e6, which == 3 ? 0.0 : (double)steps / total);
            check(name);
        };
        run(0, "B ordered u64 atomicMin, agent");
        run(3, "C key CAS + u32 atomicMin, agent");
        run(1, "D' ordered, agent, XCD-local lanes");
        run(2, "D ordered, WORKGROUP scope, XCD-local lanes");
        for (int l = 0; l < LANES; l++) CK(hipFree(tb[l]));
        CK(hipFree(d_tb));
    }
    return 0;
}
