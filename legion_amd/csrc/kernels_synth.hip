// kernels_synth.hip -- device-side generators for the synthetic workloads of BASELINE.md
// (W1: RMAT-26, Graph500 a,b,c,d = 0.57,0.19,0.19,0.05; counter-hash float32 features).
// The reference's offline dataset tooling (dataset/*, a Java WebGraph pipeline) is out of scope;
// these produce arrays in the reference's in-memory formats directly in HBM.
// Every value is a pure function of (seed, index), so a gathered feature row can be verified
// byte-for-byte at full size without any host copy of the table (legion_synth_feature_check).
#include "legion_core.h"

namespace lg {

__host__ __device__ inline uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// Graph500-style label scrambling: a bijection of [0, 2^scale) (odd multiplier, xor-shift, odd multiplier:
// every step is invertible modulo 2^scale), so hubs no longer sit at the low vertex ids and hot rows of
// node_map / RowHdr / the feature table are spread over the whole id range like in a real graph.
__host__ __device__ inline uint32_t scramble_label(uint32_t x, int32_t scale, uint64_t key)
{
    const uint32_t mask = scale >= 32 ? 0xFFFFFFFFu : ((1u << scale) - 1u);
    const uint32_t m1 = (uint32_t)(key) | 1u, m2 = (uint32_t)(key >> 32) | 1u;
    const int32_t sh = scale > 1 ? (scale + 1) / 2 : 1;
    x = (x * m1 + (uint32_t)(key >> 17)) & mask;
    x ^= x >> sh;
    x = (x * m2) & mask;
    x ^= x >> sh;
    return x & mask;
}

// one RMAT edge per thread; quadrant thresholds in 16.16 fixed point of (a, a+b, a+b+c)
__global__ void rmat_kernel(int32_t scale, int64_t num_edges, uint64_t seed, int32_t* __restrict__ src_out,
                            int32_t* __restrict__ dst_out, uint64_t scramble_key)
{
    const uint32_t ta = (uint32_t)(0.57 * 65536.0), tab = (uint32_t)(0.76 * 65536.0), tabc = (uint32_t)(0.95 * 65536.0);
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < num_edges;
         e += (int64_t)gridDim.x * blockDim.x) {
        uint32_t u = 0, v = 0;
        uint64_t h = 0;
        for (int32_t level = 0; level < scale; level++) {
            if ((level & 3) == 0) h = splitmix64(seed ^ ((uint64_t)e * 8ull + (uint64_t)(level >> 2)));
            const uint32_t r = (uint32_t)(h >> ((level & 3) * 16)) & 0xFFFFu;
            const uint32_t ubit = r >= tab ? 1u : 0u;                       // quadrants c, d
            const uint32_t vbit = (r >= ta && r < tab) || r >= tabc ? 1u : 0u;   // quadrants b, d
            u = (u << 1) | ubit;
            v = (v << 1) | vbit;
        }
        if (u == v) v = u ^ 1u;                                             // no self loops
        if (scramble_key != 0) {
            u = scramble_label(u, scale, scramble_key);
            v = scramble_label(v, scale, scramble_key);
        }
        src_out[e] = (int32_t)u;
        dst_out[e] = (int32_t)v;
    }
}

__host__ __device__ inline float synth_feature_value(uint64_t seed, int64_t row, int32_t dim, int32_t f)
{
    const uint64_t h = splitmix64(seed ^ (uint64_t)(row * (int64_t)dim + f));
    return (float)(int32_t)((uint32_t)(h >> 40)) * (1.0f / 8388608.0f) - 1.0f;   // 24 bits -> [-1, 1), exact
}

__global__ void synth_features_kernel(float* __restrict__ out, int64_t first_row, int64_t num_rows, int32_t dim,
                                      uint64_t seed)
{
    const int64_t total = num_rows * dim;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / dim;
        const int32_t f = (int32_t)(i - r * dim);
        out[i] = synth_feature_value(seed, first_row + r, dim, f);
    }
}

__global__ void synth_feature_check_kernel(const float* __restrict__ rows, const int32_t* __restrict__ ids,
                                           int64_t num_rows, int32_t dim, uint64_t seed,
                                           unsigned long long* __restrict__ mismatch)
{
    const int64_t total = num_rows * dim;
    unsigned long long bad = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / dim;
        const int32_t f = (int32_t)(i - r * dim);
        const int32_t id = ids[r];
        if (id < 0) continue;
        const float want = synth_feature_value(seed, id, dim, f);
        if (__float_as_uint(want) != __float_as_uint(rows[i])) bad++;
    }
    if (bad) atomicAdd(mismatch, bad);
}

}  // namespace lg

extern "C" void legion_synth_rmat_edges_scrambled(legion_stream_t stream, int32_t scale, int64_t num_edges, uint64_t seed,
                                                  int32_t* src_out, int32_t* dst_out, uint64_t scramble_key)
{
    if (num_edges <= 0) return;
    lg::rmat_kernel<<<4096, 256, 0, static_cast<hipStream_t>(stream)>>>(scale, num_edges, seed, src_out, dst_out,
                                                                       scramble_key);
    hipCheckError();
}

extern "C" void legion_synth_rmat_edges(legion_stream_t stream, int32_t scale, int64_t num_edges, uint64_t seed,
                                        int32_t* src_out, int32_t* dst_out)
{
    legion_synth_rmat_edges_scrambled(stream, scale, num_edges, seed, src_out, dst_out, 0);
}

extern "C" void legion_synth_features(legion_stream_t stream, float* out, int64_t first_row, int64_t num_rows,
                                      int32_t dim, uint64_t seed)
{
    if (num_rows <= 0 || dim <= 0) return;
    lg::synth_features_kernel<<<8192, 256, 0, static_cast<hipStream_t>(stream)>>>(out, first_row, num_rows, dim, seed);
    hipCheckError();
}

extern "C" void legion_synth_feature_check(legion_stream_t stream, const float* rows, const int32_t* ids,
                                           int64_t num_rows, int32_t dim, uint64_t seed,
                                           unsigned long long* mismatch_count_devptr)
{
    if (num_rows <= 0 || dim <= 0) return;
    lg::synth_feature_check_kernel<<<2048, 256, 0, static_cast<hipStream_t>(stream)>>>(rows, ids, num_rows, dim, seed,
                                                                                      mismatch_count_devptr);
    hipCheckError();
}

// ---- measurement aid: a consumer that READS what it is handed --------------------------------------------------------
// One launch per mini-batch: every float of the feature rows and every entry of the two COO arrays is loaded and folded into a
// device accumulator (what the first layer of a GNN does to a batch at the very least).  tools/server_throughput.py --consume
// launches it per get_next, so that the boundary figure is a rate with the batch actually read on the trainer's side, not a
// rate of hand-overs nobody looks at.
namespace lg {
__global__ __launch_bounds__(256) void consume_batch_kernel(const float* __restrict__ feats, int64_t n_floats, const int32_t* __restrict__ a,
                                                           const int32_t* __restrict__ b, int64_t n_edges, double* __restrict__ acc)
{
    typedef float v4 __attribute__((ext_vector_type(4)));
    const int64_t stride = (int64_t)gridDim.x * 256, t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    float s = 0.f;
    long long e = 0;
    const int64_t q = ((reinterpret_cast<uintptr_t>(feats) & 15) == 0) ? n_floats >> 2 : 0;
    for (int64_t i = t; i < q; i += stride) { const v4 v = reinterpret_cast<const v4*>(feats)[i]; s += (v.x + v.y) + (v.z + v.w); }
    for (int64_t i = (q << 2) + t; i < n_floats; i += stride) s += feats[i];
    for (int64_t i = t; i < n_edges; i += stride) e += (long long)a[i] + b[i];
    double w = (double)s + (double)e;
    for (int off = 32; off > 0; off >>= 1) w += __shfl_down(w, off);
    if ((threadIdx.x & 63) == 0 && w != 0.0) atomicAdd(acc, w);
}
}  // namespace lg

extern "C" void legion_consume_batch(legion_stream_t stream, const float* feats, int64_t n_floats, const int32_t* src, const int32_t* dst,
                                     int64_t n_edges, double* acc_devptr)
{
    if (acc_devptr == nullptr || (n_floats <= 0 && n_edges <= 0)) return;
    int64_t grid = (n_floats / 4 + n_edges + 1023) / 1024;
    if (grid > 2048) grid = 2048;
    if (grid < 1) grid = 1;
    lg::consume_batch_kernel<<<(int32_t)grid, 256, 0, static_cast<hipStream_t>(stream)>>>(feats, n_floats, src, dst, n_edges, acc_devptr);
    hipCheckError();
}
