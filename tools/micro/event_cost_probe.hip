// Probe: what do back-to-back small kernels cost on one stream with and without an event record after each,
// and with an in-kernel "last workgroup writes a host flag" completion instead?
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error '%s' at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__global__ void work(float* p, int n, unsigned* ticket, unsigned* flag, unsigned seq)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = p[i] * 1.0001f + 1.0f;
    if (flag) {
        __threadfence();
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            if (t == gridDim.x - 1) {
                __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const int n = 4 << 20, N = 400;
    float* d; CK(hipMalloc(&d, n * 4)); CK(hipMemset(d, 0, n * 4));
    unsigned* ticket; CK(hipMalloc(&ticket, 4)); CK(hipMemset(ticket, 0, 4));
    unsigned* flag_h; void* flag_d; CK(hipHostMalloc((void**)&flag_h, 64, hipHostMallocMapped)); CK(hipHostGetDevicePointer(&flag_d, flag_h, 0)); *flag_h = 0;
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t ev[2]; CK(hipEventCreateWithFlags(&ev[0], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ev[1], hipEventDisableTiming));
    for (int mode = 0; mode < 3; mode++) {
        for (int rep = 0; rep < 2; rep++) {
            CK(hipStreamSynchronize(s));
            double t0 = now();
            for (int i = 0; i < N; i++) {
                work<<<1024, 256, 0, s>>>(d, n, ticket, mode == 2 ? (unsigned*)flag_d : nullptr, (unsigned)(rep * N + i + 1));
                if (mode == 1) CK(hipEventRecord(ev[i & 1], s));
            }
            CK(hipStreamSynchronize(s));
            double t1 = now();
            if (rep == 1) printf("%s: %.2f us per kernel (16 MB in + 16 MB out, 1024 WGs), back to back\n",
                                 mode == 0 ? "no events" : mode == 1 ? "hipEventRecord after each" : "in-kernel last-WG host flag", (t1 - t0) / N * 1e6);
        }
    }
    // latency of seeing completion: event query vs host flag
    double a = 0, b = 0;
    for (int i = 0; i < 200; i++) {
        CK(hipStreamSynchronize(s));
        double t0 = now();
        work<<<1024, 256, 0, s>>>(d, n, ticket, nullptr, 0);
        CK(hipEventRecord(ev[0], s));
        while (hipEventQuery(ev[0]) == hipErrorNotReady) {}
        a += now() - t0;
        unsigned seq = 100000 + i;
        t0 = now();
        work<<<1024, 256, 0, s>>>(d, n, ticket, (unsigned*)flag_d, seq);
        while (*(volatile unsigned*)flag_h != seq) {}
        b += now() - t0;
    }
    printf("launch -> completion seen: event %.2f us, host flag %.2f us\n", a / 200 * 1e6, b / 200 * 1e6);
    return 0;
}
