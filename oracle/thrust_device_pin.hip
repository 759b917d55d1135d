// thrust_device_pin.hip -- runs the reference's exact Thrust DEVICE expressions on the MI355X.
//
// TEST INFRASTRUCTURE ONLY (built by oracle/Makefile into oracle/_build/, run by tests/test_gpu_thrust_pin.py).
// thrust_pin.cpp pins the draw and the hotness order to rocThrust's HOST code; the reference, however, ran the
// device paths.  This program executes those, with the reference's own expressions, on the GPU:
//   SS/cache/cache.cu:414-415,434-435   init_cache_order (iota) + thrust::sort_by_key(thrust::device, keys, keys + N,
//                                       order, thrust::greater<unsigned long long>())
//   SS/cache/cache.cu:471-472,500       thrust::inclusive_scan(thrust::device, in, in + N, out)
//   SS/engine/operator_impl.cu:235-238  thrust::minstd_rand engine; engine.discard(idx);
//                                       thrust::uniform_int_distribution<> dist(0, col_size - 1); dist(engine)   (in a kernel)
// and writes the results to files; the test compares them with the product's rocPRIM / table-driven path and with
// the oracle, at N >= 2^24 with massive ties (most vertices tie at hotness 0, as after a PreSC epoch).
//
//   thrust_device_pin <keys.u64> <n> <out_order.i32> <out_sorted.u64> <out_scan.u64> <pairs.i32 (idx,deg)*m> <m> <out_draws.i32>
#include <hip/hip_runtime.h>
#include <thrust/device_ptr.h>
#include <thrust/execution_policy.h>
#include <thrust/functional.h>
#include <thrust/random/linear_congruential_engine.h>
#include <thrust/random/uniform_int_distribution.h>
#include <thrust/scan.h>
#include <thrust/sort.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void init_cache_order(int32_t* cache_order, int32_t total_num_nodes)   // cache_impl.cuh:79-83
{
    for (int32_t thread_idx = threadIdx.x + blockDim.x * blockIdx.x; thread_idx < total_num_nodes; thread_idx += gridDim.x * blockDim.x)
        cache_order[thread_idx] = thread_idx;
}

__global__ void draw_kernel(const int32_t* pairs, int32_t m, int32_t* out)
{
    for (int32_t i = threadIdx.x + blockDim.x * blockIdx.x; i < m; i += gridDim.x * blockDim.x) {
        const int32_t idx = pairs[2 * i], col_size = pairs[2 * i + 1];
        thrust::minstd_rand engine;                                   // operator_impl.cu:235
        engine.discard(idx);                                          // :236
        thrust::uniform_int_distribution<> dist(0, col_size - 1);     // :237
        out[i] = dist(engine);                                        // :238
    }
}

template <typename T>
static std::vector<T> read_file(const char* path, size_t n)
{
    std::vector<T> v(n);
    FILE* f = fopen(path, "rb");
    if (!f || fread(v.data(), sizeof(T), n, f) != n) { fprintf(stderr, "cannot read %s\n", path); exit(1); }
    fclose(f);
    return v;
}

template <typename T>
static void write_file(const char* path, const T* dev, size_t n)
{
    std::vector<T> v(n);
    CK(hipMemcpy(v.data(), dev, n * sizeof(T), hipMemcpyDeviceToHost));
    FILE* f = fopen(path, "wb");
    if (!f || fwrite(v.data(), sizeof(T), n, f) != n) { fprintf(stderr, "cannot write %s\n", path); exit(1); }
    fclose(f);
}

int main(int argc, char** argv)
{
    if (argc != 9) { fprintf(stderr, "usage: see the header comment\n"); return 2; }
    const size_t n = strtoull(argv[2], nullptr, 10), m = strtoull(argv[7], nullptr, 10);
    std::vector<unsigned long long> keys = read_file<unsigned long long>(argv[1], n);
    std::vector<int32_t> pairs = read_file<int32_t>(argv[6], 2 * m);

    unsigned long long *d_keys, *d_scan;
    int32_t *d_order, *d_pairs, *d_draws;
    CK(hipMalloc(&d_keys, n * 8));
    CK(hipMalloc(&d_scan, n * 8));
    CK(hipMalloc(&d_order, n * 4));
    CK(hipMalloc(&d_pairs, 2 * m * 4));
    CK(hipMalloc(&d_draws, m * 4));
    CK(hipMemcpy(d_keys, keys.data(), n * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_pairs, pairs.data(), 2 * m * 4, hipMemcpyHostToDevice));

    init_cache_order<<<80, 1024>>>(d_order, (int32_t)n);                                       // cache.cu:414
    thrust::sort_by_key(thrust::device, d_keys, d_keys + n, d_order, thrust::greater<unsigned long long>());   // :415
    CK(hipDeviceSynchronize());
    thrust::inclusive_scan(thrust::device, d_keys, d_keys + n, d_scan);                        // :471-472 on the sorted counts
    CK(hipDeviceSynchronize());
    draw_kernel<<<1024, 256>>>(d_pairs, (int32_t)m, d_draws);
    CK(hipDeviceSynchronize());

    write_file(argv[3], d_order, n);
    write_file(argv[4], d_keys, n);
    write_file(argv[5], d_scan, n);
    write_file(argv[8], d_draws, m);
    printf("thrust device pin: n=%zu m=%zu ok (THRUST_VERSION %d)\n", n, m, THRUST_VERSION);
    return 0;
}
