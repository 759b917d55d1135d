// Micro-benchmark: rate of random 8-byte loads from an 8 GB table (the sampler's pick of {neighbour, cache slot}) by kind of allocation
// (coarse-grained hipMalloc, fine-grained, uncached) and by the load's cache-policy bits.  Question: does any of them make the memory
// system fetch less than a whole 128-byte line per pick?   hipcc --offload-arch=gfx950 -O3 random_load_policy.hip -o random_load_policy
// Run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` to see the bytes behind each variant.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef unsigned long long u64;
__device__ inline uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
template <int POL> __device__ inline u64 ld(const u64* p)
{
    u64 v;
    if (POL == 0) v = *p;
    if (POL == 1) v = __builtin_nontemporal_load(p);
    if (POL == 2) asm volatile("global_load_dwordx2 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (POL == 3) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (POL == 4) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1 nt\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (POL == 5) asm volatile("global_load_dwordx2 %0, %1, off nt\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
// (the inline-asm variants wait for each load: to keep several in flight per wave the kernel runs 4 x the waves instead -- the
// plain variant is measured both ways: POL 0 = compiler-scheduled, POL 6 = plain load through the same asm + wait)
template <> __device__ inline u64 ld<6>(const u64* p)
{
    u64 v;
    asm volatile("global_load_dwordx2 %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
template <int POL>
__global__ __launch_bounds__(256) void k(const u64* table, uint32_t mask, uint32_t n, uint32_t salt, u64* sink)
{
    u64 acc = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) acc += ld<POL>(table + (mix(i * 2654435761u + salt) & mask));
    if (acc == 0x12345678u) *sink = acc;
}
template <int POL> static float run(const u64* t, uint32_t mask, uint32_t n, u64* sink)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float ms = 0, best = 1e9f;
    for (int rep = 0; rep < 4; rep++) {
        hipEventRecord(a);
        k<POL><<<16384, 256>>>(t, mask, n, 17u + rep, sink);
        hipEventRecord(b); hipEventSynchronize(b);
        hipEventElapsedTime(&ms, a, b);
        if (rep > 0 && ms < best) best = ms;
    }
    return best;
}
int main()
{
    const uint32_t n = 64u << 20;          // 64 M loads per launch
    const size_t bytes = 8ull << 30;
    const uint32_t mask = (uint32_t)(bytes / 8 - 1);
    u64* sink; hipMalloc(&sink, 8);
    const char* kinds[3] = {"hipMalloc (coarse-grained)", "fine-grained", "uncached"};
    for (int kind = 0; kind < 3; kind++) {
        u64* t = nullptr;
        hipError_t e = kind == 0 ? hipMalloc(&t, bytes)
                     : hipExtMallocWithFlags((void**)&t, bytes, kind == 1 ? hipDeviceMallocFinegrained : hipDeviceMallocUncached);
        if (e != hipSuccess) { printf("%s: allocation failed (%s)\n", kinds[kind], hipGetErrorString(e)); (void)hipGetLastError(); continue; }
        hipMemset(t, 1, bytes);
        hipDeviceSynchronize();
        const float m0 = run<0>(t, mask, n, sink), m1 = run<1>(t, mask, n, sink), m6 = run<6>(t, mask, n, sink), m2 = run<2>(t, mask, n, sink),
                    m3 = run<3>(t, mask, n, sink), m4 = run<4>(t, mask, n, sink), m5 = run<5>(t, mask, n, sink);
        auto g = [&](float ms) { return n / ms / 1e6; };
        printf("%-28s G loads/s: plain %5.1f  nontemporal(builtin) %5.1f | asm+wait: plain %5.1f  sc1 %5.1f  sc0 sc1 %5.1f  sc0 sc1 nt %5.1f  nt %5.1f\n",
               kinds[kind], g(m0), g(m1), g(m6), g(m2), g(m3), g(m4), g(m5));
        fflush(stdout);
        hipFree(t);
    }
    return 0;
}
