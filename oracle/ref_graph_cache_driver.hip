// ref_graph_cache_driver.hip -- runs the REFERENCE'S OWN topology-cache kernels on the GPU.
//
// TEST INFRASTRUCTURE ONLY (oracle/: the checker, never the product).  The reference is CUDA + libcu++ + inline PTX and cannot be
// built in this image -- with one exception: sampling_server/src/storage/graph_storage_impl.cuh, the file that holds the
// kernels of GraphStorage::GraphCache (GetNeighborCount :33-39, TopoFillUp :41-53), includes nothing but the standard library and
// Thrust (rocThrust here) and is plain kernel code.  oracle/Makefile target `ref` compiles THAT FILE, where it lies under
// /root/reference, with hipcc into oracle/_ref/ref_graph_cache (no copy of it enters this repository, no stand-in header is written:
// LEGION_REF_GRAPH_STORAGE_IMPL is its path, given on the compiler's command line).  The host wrapper that launches the two kernels,
// GraphStorage::GraphCache (SS/storage/graph_storage.cu:76-111), is CUDA-API host code inside a file that does not compile here; this
// driver restates its call sequence (allocate, GetNeighborCount, thrust::inclusive_scan, TopoFillUp: :84-100) with HIP calls, grid
// shapes included, around the reference's kernels.
//
//   ref_graph_cache <in> <out>
//   in : int64 {N, E, Kg, capacity} | int32 QT[N] | int64 indptr[N+1] | int32 col[E]
//   out: for i = 0 .. Kg-1: int64 index[capacity+1] | int64 n | int32 dst[n]         (GPU i of the clique caches QT[r*Kg + i])
// tests/test_gpu_ref_graph_cache.py compares the output with the oracle's restatement (lgo_fill_up) and with the product's cached CSR.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include <thrust/execution_policy.h>
#include <thrust/scan.h>

#include LEGION_REF_GRAPH_STORAGE_IMPL

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP failure %s:%d: '%s'\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

template <typename T>
static std::vector<T> read_n(FILE* f, size_t n)
{
    std::vector<T> v(n);
    if (n && fread(v.data(), sizeof(T), n, f) != n) { printf("short input\n"); exit(1); }
    return v;
}

int main(int argc, char** argv)
{
    if (argc != 3) { printf("usage: %s <in> <out>\n", argv[0]); return 2; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { printf("cannot open %s\n", argv[1]); return 1; }
    const std::vector<int64_t> hdr = read_n<int64_t>(f, 4);
    const int64_t N = hdr[0], E = hdr[1];
    const int32_t Kg = (int32_t)hdr[2], capacity = (int32_t)hdr[3];
    if ((int64_t)capacity * Kg > N) { printf("capacity * Kg > N: the reference would read QT out of bounds\n"); return 1; }
    const std::vector<int32_t> QT = read_n<int32_t>(f, (size_t)N);
    const std::vector<int64_t> indptr = read_n<int64_t>(f, (size_t)N + 1);
    const std::vector<int32_t> col = read_n<int32_t>(f, (size_t)E);
    fclose(f);
    int32_t *d_QT, *d_col;
    int64_t* d_indptr;
    CK(hipMalloc(&d_QT, (size_t)N * 4));
    CK(hipMalloc(&d_indptr, ((size_t)N + 1) * 8));
    CK(hipMalloc(&d_col, (size_t)(E > 0 ? E : 1) * 4));
    CK(hipMemcpy(d_QT, QT.data(), (size_t)N * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_indptr, indptr.data(), ((size_t)N + 1) * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_col, col.data(), (size_t)E * 4, hipMemcpyHostToDevice));
    FILE* o = fopen(argv[2], "wb");
    if (!o) { printf("cannot open %s\n", argv[2]); return 1; }
    for (int32_t i = 0; i < Kg; i++) {                                                    // graph_storage.cu:81
        int64_t* neighbor_count;
        CK(hipMalloc(&neighbor_count, (size_t)(capacity > 0 ? capacity : 1) * sizeof(int64_t)));            // :84
        GetNeighborCount<<<128, 1024>>>(d_QT, Kg, i, capacity, d_indptr, neighbor_count);                   // :85
        int64_t* d_csr_node_index;
        CK(hipMalloc(&d_csr_node_index, ((size_t)capacity + 1) * sizeof(int64_t)));                         // :87-88
        CK(hipMemset(d_csr_node_index, 0, ((size_t)capacity + 1) * sizeof(int64_t)));                       // :89
        thrust::inclusive_scan(thrust::device, neighbor_count, neighbor_count + capacity, d_csr_node_index + 1);   // :90
        CK(hipGetLastError());
        std::vector<int64_t> h_csr_node_index((size_t)capacity + 1);
        CK(hipMemcpy(h_csr_node_index.data(), d_csr_node_index, ((size_t)capacity + 1) * sizeof(int64_t), hipMemcpyDeviceToHost));   // :92-93
        int32_t* d_csr_dst_node_ids;
        const int64_t n = h_csr_node_index[capacity];
        CK(hipMalloc(&d_csr_dst_node_ids, (size_t)(n > 0 ? n : 1) * sizeof(int32_t)));                      // :95-96
        TopoFillUp<<<80, 1024>>>(d_QT, Kg, i, capacity, d_indptr, d_col, d_csr_node_index, d_csr_dst_node_ids);     // :98
        CK(hipGetLastError());
        CK(hipDeviceSynchronize());
        std::vector<int32_t> h_dst((size_t)n);
        CK(hipMemcpy(h_dst.data(), d_csr_dst_node_ids, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost));
        fwrite(h_csr_node_index.data(), sizeof(int64_t), h_csr_node_index.size(), o);
        fwrite(&n, sizeof(int64_t), 1, o);
        fwrite(h_dst.data(), sizeof(int32_t), h_dst.size(), o);
        CK(hipFree(neighbor_count));                                                                        // :102
        CK(hipFree(d_csr_node_index));
        CK(hipFree(d_csr_dst_node_ids));
    }
    fclose(o);
    return 0;
}
