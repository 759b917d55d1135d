// Probe: can a queued kernel wait on a host flag at the command processor (hipStreamWaitValue32) and report its
// completion through hipStreamWriteValue32, and what does a release -> done round trip cost compared with
// launch + hipEventRecord + hipEventQuery?   hipcc --offload-arch=gfx950 -O2 stream_wait_probe.hip -o stream_wait_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error '%s' at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__global__ void work(float* p, int n) { for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = p[i] * 1.0001f + 1.0f; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const int n = 4 << 20;   // 16 MB read + 16 MB written, like a B = 1024 hand-over
    float* d; CK(hipMalloc(&d, n * 4)); CK(hipMemset(d, 0, n * 4));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    // (1) launch + event + query
    double t_launch = 0, t_total = 0;
    for (int it = 0; it < 200; it++) {
        double t0 = now();
        work<<<580, 256, 0, s>>>(d, n);
        CK(hipEventRecord(ev, s));
        double t1 = now();
        while (hipEventQuery(ev) == hipErrorNotReady) {}
        double t2 = now();
        if (it >= 20) { t_launch += t1 - t0; t_total += t2 - t0; }
    }
    printf("launch+event+query: API %.1f us, launch -> completion seen %.1f us\n", t_launch / 180 * 1e6, t_total / 180 * 1e6);
    // (2) pre-queued kernel waiting on a host flag, completion through a stream write
    uint32_t *flag_h, *done_h; void *flag_d, *done_d;
    if (hipExtMallocWithFlags((void**)&flag_h, 8, hipMallocSignalMemory) != hipSuccess) { printf("no signal memory\n"); return 0; }
    CK(hipHostMalloc((void**)&done_h, 64, hipHostMallocMapped));
    flag_d = flag_h;
    CK(hipHostGetDevicePointer(&done_d, done_h, 0));
    *flag_h = 0; *done_h = 0;
    double t_rt = 0;
    for (uint32_t it = 1; it <= 200; it++) {
        hipError_t e = hipStreamWaitValue32(s, flag_d, it, hipStreamWaitValueGte, 0xFFFFFFFFu);
        if (e != hipSuccess) { printf("hipStreamWaitValue32: %s\n", hipGetErrorString(e)); return 0; }
        work<<<580, 256, 0, s>>>(d, n);
        e = hipStreamWriteValue32(s, done_d, it, 0);
        if (e != hipSuccess) { printf("hipStreamWriteValue32: %s\n", hipGetErrorString(e)); return 0; }
        std::this_thread::sleep_for(std::chrono::microseconds(200));      // everything is queued and waiting by now
        double t0 = now();
        *(volatile uint32_t*)flag_h = it;                                  // "the trainer released the slot"
        double tmax = t0 + 2.0;
        while (*(volatile uint32_t*)done_h < it) { if (now() > tmax) { printf("timeout waiting for done (it %u)\n", it); *(volatile uint32_t*)flag_h = 0xFFFFFFF0u; CK(hipStreamSynchronize(s)); return 0; } }
        double t1 = now();
        if (it > 20) t_rt += t1 - t0;
    }
    printf("pre-queued (wait value -> kernel -> write value): release -> completion seen %.1f us\n", t_rt / 180 * 1e6);
    CK(hipStreamSynchronize(s));
    return 0;
}
