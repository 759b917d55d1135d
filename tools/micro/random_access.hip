// Micro-benchmark: rate of random 4-byte atomicMin / loads / stores into tables of different sizes
// (L2-, Infinity-Cache- and HBM-resident) on MI355X.  hipcc --offload-arch=gfx950 -O3 random_access.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__device__ inline uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
template <int MODE>
__global__ void k(uint32_t* table, uint32_t mask, uint32_t n, uint32_t salt, uint32_t* sink)
{
    uint32_t acc = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        uint32_t a = mix(i * 2654435761u + salt) & mask;
        if (MODE == 0) atomicMin(table + a, i);
        if (MODE == 1) acc += table[a];
        if (MODE == 2) table[a] = i;
    }
    if (MODE == 1 && acc == 0x12345678u) *sink = acc;
}
int main()
{
    const uint32_t n = 2u << 20;   // 2 M accesses per launch
    uint32_t* sink; hipMalloc(&sink, 4);
    for (size_t mb : {1, 4, 16, 64, 128, 256, 512, 1024, 4096}) {
        size_t bytes = mb << 20;
        uint32_t* t; hipMalloc(&t, bytes); hipMemset(t, 0xFF, bytes);
        uint32_t mask = (uint32_t)(bytes / 4 - 1);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        float ms[3];
        for (int mode = 0; mode < 3; mode++) {
            for (int rep = 0; rep < 3; rep++) {   // same addresses every rep: warm caches where they fit
                hipEventRecord(a);
                if (mode == 0) k<0><<<2048, 256>>>(t, mask, n, 1, sink);
                if (mode == 1) k<1><<<2048, 256>>>(t, mask, n, 1, sink);
                if (mode == 2) k<2><<<2048, 256>>>(t, mask, n, 1, sink);
                hipEventRecord(b); hipEventSynchronize(b);
                hipEventElapsedTime(&ms[mode], a, b);
            }
        }
        printf("table %5zu MB: atomicMin %6.1f us (%5.1f G/s)  load %6.1f us (%5.1f G/s)  store %6.1f us (%5.1f G/s)\n", mb,
               ms[0] * 1e3, n / ms[0] / 1e6, ms[1] * 1e3, n / ms[1] / 1e6, ms[2] * 1e3, n / ms[2] / 1e6);
        hipFree(t);
    }
    return 0;
}
