// collective.hip -- the one collective of the path: the sum of the per-GPU access counters (hotness) over the GPUs of a
// clique, as an RCCL all-reduce over xGMI.
//
// Reference: SS/cache/cache.cu:408-411,428-431 -- `aggregate_access<<<80,1024>>>` on the clique leader, reading every
// member's uint64[N] node / edge counters through peer pointers (there is no collective library in the reference's
// server).  At RMAT-28 that is one GPU pulling 7 x 2 x 2.15 GB over its own seven links while seven GPUs
// idle.  BASELINE.json's north_star assigns this step to RCCL, and this file is where the product issues it:
//
//   * one server process, one host thread per GPU (the reference's deployment; GPUServer / UnifiedCache::
//     CandidateSelection): ncclCommInitAll over the clique's DISTINCT physical GPUs, one
//     ncclAllReduce(ncclUint64, ncclSum) per array and member, in place, inside a group call -- afterwards every member
//     holds the clique sum and the leader sorts its own copy.  Message: N x 8 bytes per array and GPU (RMAT-26: 537 MB,
//     RMAT-28: 2.15 GB, uk-union: 1.07 GB), twice (node and edge hotness).  RCCL on a fully connected xGMI mesh runs this
//     as a ring / direct reduce-scatter + all-gather moving 2 (K-1)/K x N x 8 bytes per GPU and array: at ~153 GB/s per
//     link direction and 7 links busy, 2 x 1.75 x 2.15 GB / (7 x 153 GB/s) = ~7 ms per array at RMAT-28 with 8 GPUs in the
//     ideal case, ~50 ms if a ring is bound by one link (2 x 7/8 x 2.15 GB / 153 GB/s = 25 ms per array); the leader loop it
//     replaces moves 7 x 2.15 GB INTO one GPU per array: 15 GB / (7 x 153 GB/s) = 14 ms at best, serialised behind 7 kernels.
//   * one process per GPU (bench.py under torchrun): a process-wide communicator created from a unique id that the host
//     program carries between the ranks (legion_collective_unique_id / legion_collective_init_rank -- the carrier can be
//     anything: a file, MPI, torch.distributed's store), then legion_cache_allreduce_hotness.
//
// Logical GPUs that share a physical device (tests on a 1-GPU box) cannot form a communicator (RCCL refuses duplicate
// devices): those cliques keep the leader loop, and the log says which path ran.
#include "legion_core.h"

#include <rccl/rccl.h>

#include <chrono>
#include <cstring>
#include <iostream>
#include <map>
#include <mutex>

#define RCCL_CALL(expr)                                                                       \
    {                                                                                         \
        ncclResult_t r_ = (expr);                                                             \
        if (r_ != ncclSuccess) {                                                              \
            printf("RCCL failure %s:%d: '%s'\n", __FILE__, __LINE__, ncclGetErrorString(r_)); \
            exit(EXIT_FAILURE);                                                               \
        }                                                                                     \
    }

namespace {
struct CliqueComm {
    std::vector<ncclComm_t> comms;
    std::vector<hipStream_t> streams;
};
std::mutex g_mu;
std::map<std::vector<int>, CliqueComm> g_cliques;     // key: the members' physical devices, in member order

// the process-wide communicator of the one-process-per-GPU layout
ncclComm_t g_rank_comm = nullptr;
hipStream_t g_rank_stream = nullptr;
int32_t g_rank_world = 0, g_rank_dev = -1;
}  // namespace

int lg_physical_device(int32_t dev);     // storage.hip

namespace lg {

// true when the logical GPUs `devs` sit on pairwise distinct physical devices
bool clique_is_physical(const std::vector<int32_t>& devs)
{
    std::vector<int> phys;
    for (int32_t d : devs) {
        const int p = lg_physical_device(d);
        for (int q : phys)
            if (q == p) return false;
        phys.push_back(p);
    }
    return true;
}

// All-reduce (sum) of `count` uint64 over the members of a clique that live in THIS process: send[j] is member devs[j]'s array
// on its own GPU, recv[j] where that member receives the sum (recv[j] == send[j]: in place).  Called from one host thread (group
// call).  Returns the milliseconds it took, or -1 when RCCL refused (communicator or call): the caller decides whether another
// way to form the sum exists (UnifiedCache::CandidateSelection falls back to the reference's leader loop in `auto`).
double allreduce_u64_clique(const std::vector<int32_t>& devs, const std::vector<unsigned long long*>& send,
                            const std::vector<unsigned long long*>& recv, int64_t count)
{
#define RCCL_SOFT(expr)                                                                       \
    {                                                                                         \
        ncclResult_t r_ = (expr);                                                             \
        if (r_ != ncclSuccess) {                                                              \
            printf("RCCL failure %s:%d: '%s'\n", __FILE__, __LINE__, ncclGetErrorString(r_)); \
            fflush(stdout);                                                                   \
            return -1.0;                                                                      \
        }                                                                                     \
    }
    std::vector<int> phys;
    for (int32_t d : devs) phys.push_back(lg_physical_device(d));
    std::lock_guard<std::mutex> lk(g_mu);
    CliqueComm& cc = g_cliques[phys];
    if (cc.comms.empty()) {
        std::vector<ncclComm_t> comms(phys.size());
        RCCL_SOFT(ncclCommInitAll(comms.data(), (int)phys.size(), phys.data()));
        cc.comms = comms;
        cc.streams.resize(phys.size());
        for (size_t j = 0; j < phys.size(); j++) {
            SetGPUDevice(devs[j]);
            HIP_CALL(hipStreamCreateWithFlags(&cc.streams[j], hipStreamNonBlocking));
        }
    }
    for (size_t j = 0; j < devs.size(); j++) {        // whatever filled the counters ran on other streams
        SetGPUDevice(devs[j]);
        HIP_CALL(hipDeviceSynchronize());
    }
    const auto t0 = std::chrono::steady_clock::now();
    RCCL_SOFT(ncclGroupStart());
    for (size_t j = 0; j < devs.size(); j++) {
        SetGPUDevice(devs[j]);
        RCCL_SOFT(ncclAllReduce(send[j], recv[j], (size_t)count, ncclUint64, ncclSum, cc.comms[j], cc.streams[j]));
    }
    RCCL_SOFT(ncclGroupEnd());
    for (size_t j = 0; j < devs.size(); j++) {
        SetGPUDevice(devs[j]);
        HIP_CALL(hipStreamSynchronize(cc.streams[j]));
    }
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
#undef RCCL_SOFT
}

}  // namespace lg

// ---- one process per GPU ----------------------------------------------------------------------------------------
// rank 0 calls legion_collective_unique_id and hands the 128 bytes to every rank (any carrier); every rank then calls
// legion_collective_init_rank with its rank and the logical GPU it owns.  Returns 1 on success.
// (These entry points report a failure -- message + return 0 -- instead of ending the process: a host program that has another
// way to sum the counters may fall back to it.)
#define RCCL_TRY(expr)                                                                        \
    {                                                                                         \
        ncclResult_t r_ = (expr);                                                             \
        if (r_ != ncclSuccess) {                                                              \
            printf("RCCL failure %s:%d: '%s'\n", __FILE__, __LINE__, ncclGetErrorString(r_)); \
            return 0;                                                                         \
        }                                                                                     \
    }
extern "C" int32_t legion_collective_unique_id(void* out128)
{
    if (!out128) return 0;
    ncclUniqueId id;
    RCCL_TRY(ncclGetUniqueId(&id));
    memcpy(out128, &id, sizeof(id));
    return 1;
}

extern "C" int32_t legion_collective_init_rank(const void* id128, int32_t world, int32_t rank, int32_t dev_id)
{
    if (!id128 || world < 1 || rank < 0 || rank >= world) { printf("invalid collective arguments\n"); return 0; }
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_rank_comm != nullptr) {
        RCCL_TRY(ncclCommDestroy(g_rank_comm));
        g_rank_comm = nullptr;
    }
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    SetGPUDevice(dev_id);
    g_rank_comm = nullptr;
    RCCL_TRY(ncclCommInitRank(&g_rank_comm, world, id, rank));
    if (g_rank_stream == nullptr) HIP_CALL(hipStreamCreateWithFlags(&g_rank_stream, hipStreamNonBlocking));
    g_rank_world = world;
    g_rank_dev = dev_id;
    return 1;
}

extern "C" void legion_collective_destroy(void)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_rank_comm != nullptr) RCCL_CALL(ncclCommDestroy(g_rank_comm));
    g_rank_comm = nullptr;
    g_rank_world = 0;
}

// In-place all-reduce of `count` uint64 at `devptr` over the process-wide communicator; returns the world size (0: no
// communicator).  *ms_out = wall time including the stream synchronisation.
extern "C" int32_t legion_collective_allreduce_u64(void* devptr, int64_t count, double* ms_out)
{
    if (g_rank_comm == nullptr) { printf("legion_hip: no communicator (legion_collective_init_rank)\n"); return 0; }
    SetGPUDevice(g_rank_dev);
    HIP_CALL(hipDeviceSynchronize());
    const auto t0 = std::chrono::steady_clock::now();
    RCCL_TRY(ncclAllReduce(devptr, devptr, (size_t)count, ncclUint64, ncclSum, g_rank_comm, g_rank_stream));
    HIP_CALL(hipStreamSynchronize(g_rank_stream));
    if (ms_out) *ms_out = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return g_rank_world;
}
