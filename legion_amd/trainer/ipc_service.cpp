// ipc_service.cpp -- trainer end of the server <-> trainer boundary: the PyTorch extension module
// `ipc_service` that Legion's training scripts import (training_backend/legion_graphsage.py:72-128).
//
// Reference: training_backend/ipc_service.cpp:93-100 (module + function names),
// training_backend/ipc_cuda_kernel.cu:35-235 (GPUIPCEnv, cuda_get_next),
// training_backend/helper_multiprocess.cpp (sharedMemoryCreate).  Same module name, same six
// functions, same return shapes/dtypes, same shm slab / semaphore names / IPC-handle slot order
// (0 ids, 1 features, 2 labels, 3 agg_src, 4 agg_dst, 5 node_counter, 6 edge_counter).
// HIP runtime calls replace the CUDA ones (hipIpcOpenMemHandle, hipMemcpy); there is no device
// code in this module.  LEGION_IPC_NAMESPACE (optional) must match the server's.
#include <sys/socket.h>
#include <sys/un.h>
#include <cstddef>
#include <fcntl.h>
#include <semaphore.h>
#include <sys/mman.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <string>
#include <vector>

#include <hip/hip_runtime_api.h>
#include <torch/extension.h>

#define INTRABATCH_CON 3
#define INTERBATCH_CON 2
#define MAX_DEVICE 8
#define MEMORY_USAGE 7

#define hipCheckError()                                                                       \
    {                                                                                         \
        hipError_t e_ = hipGetLastError();                                                    \
        if (e_ != hipSuccess) {                                                               \
            printf("HIP failure %s:%d: '%s'\n", __FILE__, __LINE__, hipGetErrorString(e_));   \
            exit(EXIT_FAILURE);                                                               \
        }                                                                                     \
    }

// The slab is the reference's (TB/ipc_cuda_kernel.cu:30-33), same size.  This build's server also publishes, in a shm
// object of its own ("legionIPCext<suffix>"), the counters of the batch in every (device, pipe slot) in host memory,
// written by the GPU before the batch is posted: when that object exists and its magic says so, get_next reads them
// there instead of making the reference's two blocking 64-byte device-to-host copies per batch.
#define LEGION_SHM_EXT_MAGIC 0x4C47494F
typedef struct shmStruct_st {
    int32_t steps[3];
    hipIpcMemHandle_t memHandle[MAX_DEVICE][INTERBATCH_CON][MEMORY_USAGE];
} shmStruct;
// Version 2 of that object adds the direct-view hand-over (server: ipc_env.hip): the server's mini-batches are produced into
// per-lane buffers that all live in one device allocation per GPU, the lane arena, whose IPC handle is published here.  This
// module opens it, says so in trainer_direct[dev] before its first sem_post, and from then on get_next returns VIEWS of the
// batch's lane (view[dev][pipe] = byte offsets into the arena) instead of views of the two pipe slots' buffers: the server
// then does no per-batch GPU work at all.  Same semaphores, same two-slot order, same tensors as far as a training script
// can tell.  LEGION_NO_DIRECT_VIEWS=1 keeps this module on the slot buffers.
typedef struct shmExt_st {
    int32_t ext_magic;
    int32_t ext_version;
    int32_t server_state;          // 1: the server stopped on an error
    int32_t ext_reserved;
    int32_t counters[MAX_DEVICE][INTERBATCH_CON][32];
    hipIpcMemHandle_t arena[MAX_DEVICE];
    int64_t arena_bytes[MAX_DEVICE];
    int32_t trainer_direct[MAX_DEVICE];
    int32_t view_on[MAX_DEVICE][INTERBATCH_CON];
    int64_t view[MAX_DEVICE][INTERBATCH_CON][5];
    // version 3 (ipc_env.hip): arena_kind 1 = the arena consists of arena_chunks physical chunks of arena_chunk_bytes, received as file
    // descriptors from the abstract unix socket "legion_arena_<arena_sock_pid>_<dev>" and mapped back to back
    int32_t arena_kind[MAX_DEVICE];
    int32_t arena_chunks[MAX_DEVICE];
    int64_t arena_chunk_bytes[MAX_DEVICE];
    int32_t arena_sock_pid;
    int32_t ext_reserved2;
} shmExt;
static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size is part of the wire format");

// What this module reads from the environment, once, in initialize() (it is an extension of its own: LegionTuning lives in the
// server's library).  Deployment: LEGION_IPC_NAMESPACE (suffix of every shm / semaphore name; must match the server's),
// LEGION_IPC_DEVICE (attach to this LOGICAL server GPU while staying on the current physical device).  Behaviour:
// LEGION_NO_SHM_MIRROR=1 (counters copied from the device, as the reference's trainer end does), LEGION_NO_DIRECT_VIEWS=1
// (batches from the two pipe slots only, as a build of TB/ipc_cuda_kernel.cu gets them).
struct TrainerOptions {
    std::string suffix;
    int ipc_device = -1;
    bool shm_mirror = true, direct_views = true;
    static TrainerOptions FromEnv()
    {
        TrainerOptions o;
        if (const char* e = getenv("LEGION_IPC_NAMESPACE")) o.suffix = e;
        if (const char* e = getenv("LEGION_IPC_DEVICE")) o.ipc_device = atoi(e);
        o.shm_mirror = getenv("LEGION_NO_SHM_MIRROR") == nullptr;
        o.direct_views = getenv("LEGION_NO_DIRECT_VIEWS") == nullptr;
        return o;
    }
};

#include "vmm_probe.h"      // vmm_import_fd(): imports a received descriptor whichever way THIS process's HIP runtime takes one

// The server's lane arena as chunks: connect, receive the descriptors (64 per message), import and map them in order.  An import
// handle is released as soon as its chunk is mapped (the mapping keeps the memory alive, nothing else has to); on any failure
// what was mapped is unmapped again and the range given back, and the caller falls back to the pipe slots.
static void unmap_arena_chunks(void* base, int n_mapped, int n_chunks, long long chunk_bytes)
{
    if (base == nullptr) return;
    for (int i = 0; i < n_mapped; i++)
        if (hipMemUnmap((char*)base + (size_t)i * (size_t)chunk_bytes, (size_t)chunk_bytes) != hipSuccess) (void)hipGetLastError();
    if (hipMemAddressFree(base, (size_t)n_chunks * (size_t)chunk_bytes) != hipSuccess) (void)hipGetLastError();
}
static void* map_arena_chunks(const std::string& suffix, int pid, int dev, int n_chunks, long long chunk_bytes, int hip_dev, std::string* why)
{
    const int c = socket(AF_UNIX, SOCK_STREAM | SOCK_CLOEXEC, 0);
    if (c < 0) { *why = "socket()"; return nullptr; }
    sockaddr_un addr;
    memset(&addr, 0, sizeof(addr));
    addr.sun_family = AF_UNIX;
    const int len = snprintf(addr.sun_path + 1, sizeof(addr.sun_path) - 1, "legion_arena_%d_%d", pid, dev);
    if (connect(c, (sockaddr*)&addr, (socklen_t)(offsetof(sockaddr_un, sun_path) + 1 + len)) != 0) { close(c); *why = "connect()"; return nullptr; }
    {   // whoever answers under that name hands out descriptors this process will map: it must be the same user's process
        ucred cr;
        socklen_t cl = sizeof(cr);
        if (getsockopt(c, SOL_SOCKET, SO_PEERCRED, &cr, &cl) != 0 || cr.uid != geteuid()) { close(c); *why = "the arena socket belongs to another user"; return nullptr; }
    }
    (void)suffix;
    void* base = nullptr;
    if (hipMemAddressReserve(&base, (size_t)n_chunks * (size_t)chunk_bytes, 0, nullptr, 0) != hipSuccess) { (void)hipGetLastError(); close(c); *why = "hipMemAddressReserve"; return nullptr; }
    int got = 0;
    bool ok = true;
    while (got < n_chunks && ok) {
        char payload = 0;
        iovec io = {&payload, 1};
        alignas(cmsghdr) char ctl[CMSG_SPACE(sizeof(int) * 64)];
        msghdr msg;
        memset(&msg, 0, sizeof(msg));
        msg.msg_iov = &io; msg.msg_iovlen = 1; msg.msg_control = ctl; msg.msg_controllen = sizeof(ctl);
        if (recvmsg(c, &msg, MSG_CMSG_CLOEXEC) != 1) { *why = "recvmsg()"; ok = false; break; }
        cmsghdr* cm = CMSG_FIRSTHDR(&msg);
        if (cm == nullptr || cm->cmsg_level != SOL_SOCKET || cm->cmsg_type != SCM_RIGHTS) { *why = "no descriptors in the message"; ok = false; break; }
        const int n = (int)((cm->cmsg_len - CMSG_LEN(0)) / sizeof(int));
        int fds[64];
        memcpy(fds, CMSG_DATA(cm), sizeof(int) * (size_t)n);
        for (int i = 0; i < n; i++) {
            if (ok && got < n_chunks) {
                hipMemGenericAllocationHandle_t h;
                if (vmm_import_fd(&h, fds[i]) != hipSuccess) {
                    (void)hipGetLastError();
                    *why = "hipMemImportFromShareableHandle";
                    ok = false;
                } else {
                    if (hipMemMap((char*)base + (size_t)got * (size_t)chunk_bytes, (size_t)chunk_bytes, 0, h, 0) != hipSuccess) {
                        (void)hipGetLastError();
                        *why = "hipMemMap";
                        ok = false;
                    } else {
                        got++;
                    }
                    if (hipMemRelease(h) != hipSuccess) (void)hipGetLastError();      // (mapped or not: the handle is not needed again)
                }
            }
            close(fds[i]);
        }
    }
    close(c);
    if (ok) {
        hipMemAccessDesc acc = {};
        acc.location.type = hipMemLocationTypeDevice;
        acc.location.id = hip_dev;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        if (hipMemSetAccess(base, (size_t)n_chunks * (size_t)chunk_bytes, &acc, 1) != hipSuccess) { (void)hipGetLastError(); *why = "hipMemSetAccess"; ok = false; }
    }
    if (!ok) {
        unmap_arena_chunks(base, got, n_chunks, chunk_bytes);
        return nullptr;
    }
    return base;
}

class GPUIPCEnv {
public:
    int Initialize()
    {
        opt_ = TrainerOptions::FromEnv();
        int central_device = -1;
        hipGetDevice(&central_device);
        hipCheckError();
        const int physical_device = central_device;
        // LEGION_IPC_DEVICE (testing): attach to the buffers / semaphores of this LOGICAL server GPU while staying on the
        // current physical device (a server with more logical GPUs than the box has, see storage.hip SetGPUDevice)
        if (opt_.ipc_device >= 0) central_device = opt_.ipc_device;
        const std::string shm_name = std::string("simpleIPCshm") + opt_.suffix;
        int fd = shm_open(shm_name.c_str(), O_RDWR | O_CREAT, 0777);
        if (fd < 0 || ftruncate(fd, sizeof(shmStruct)) != 0) {     // the reference's call; the slab has exactly this size
            printf("Failed to create shared memory slab\n");
            exit(EXIT_FAILURE);
        }
        void* addr = mmap(0, sizeof(shmStruct), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        if (addr == MAP_FAILED) {
            printf("Failed to create shared memory slab\n");
            exit(EXIT_FAILURE);
        }
        volatile shmStruct* shm = (volatile shmStruct*)addr;
        train_step_ = shm->steps[0];
        valid_step_ = shm->steps[1];
        test_step_ = shm->steps[2];
        std::vector<void*>* slots[MEMORY_USAGE] = {&ids_, &float_features_, &labels_, &agg_src_, &agg_dst_,
                                                   &node_counter_, &edge_counter_};
        for (auto* s : slots) s->assign(INTERBATCH_CON, nullptr);
        for (int i = 0; i < INTERBATCH_CON; i++) {
            for (int k = 0; k < MEMORY_USAGE; k++) {
                hipIpcMemHandle_t h = *(hipIpcMemHandle_t*)&shm->memHandle[central_device][i][k];
                hipError_t e = hipIpcOpenMemHandle(&(*slots[k])[i], h, hipIpcMemLazyEnablePeerAccess);
                // Observed on ROCm 7.2 (dmabuf IPC): while the exporting server process is busy allocating (its runners
                // start right after "ready") and another trainer attaches at the same moment, an open can fail transiently
                // with 'invalid device pointer'; the handle itself is fine.  Retry briefly before giving up.
                for (int attempt = 0; e != hipSuccess && attempt < 10; attempt++) {
                    (void)hipGetLastError();
                    usleep(25000);
                    e = hipIpcOpenMemHandle(&(*slots[k])[i], h, hipIpcMemLazyEnablePeerAccess);
                    if (e == hipSuccess) printf("ipc_service: IPC handle (slot %d pipe %d) opened after %d retries\n", k, i, attempt + 1);
                }
                if (e != hipSuccess) {
                    printf("HIP failure %s:%d: '%s' (slot %d pipe %d)\n", __FILE__, __LINE__, hipGetErrorString(e), k, i);
                    exit(EXIT_FAILURE);
                }
            }
        }
        std::cout << "HIP: " << central_device << " IPC shared memory opened\n";
        if (opt_.shm_mirror) {                                  // this build's server: counters in host memory (never created here)
            const std::string ext_name = std::string("legionIPCext") + opt_.suffix;
            const int efd = shm_open(ext_name.c_str(), O_RDWR, 0);
            if (efd >= 0) {
                void* ea = mmap(0, sizeof(shmExt), PROT_READ | PROT_WRITE, MAP_SHARED, efd, 0);
                if (ea != MAP_FAILED && ((volatile shmExt*)ea)->ext_magic == LEGION_SHM_EXT_MAGIC) mirror_ = (volatile shmExt*)ea;
                else if (ea != MAP_FAILED) munmap(ea, sizeof(shmExt));
                close(efd);
            }
        }
        // direct views: open the server's lane arena and say so BEFORE the first sem_post below (the server reads the flag
        // when its first wait returns)
        if (mirror_ != nullptr && mirror_->ext_version >= 3 && mirror_->arena_bytes[central_device] > 0 && mirror_->arena_kind[central_device] == 1 &&
            opt_.direct_views) {
            int hip_dev = 0;
            (void)hipGetDevice(&hip_dev);
            std::string why;
            arena_ = map_arena_chunks(opt_.suffix, mirror_->arena_sock_pid, central_device, mirror_->arena_chunks[central_device],
                                      mirror_->arena_chunk_bytes[central_device], hip_dev, &why);
            if (arena_ != nullptr) {
                arena_bytes_ = mirror_->arena_bytes[central_device];
                arena_chunks_ = mirror_->arena_chunks[central_device];
                arena_chunk_bytes_ = mirror_->arena_chunk_bytes[central_device];
                mirror_->trainer_direct[central_device] = 1;
                __sync_synchronize();
                std::cout << "HIP: " << central_device << " lane arena mapped (" << (arena_bytes_ >> 20) << " MiB in " << mirror_->arena_chunks[central_device]
                          << " chunks): batches arrive as views" << std::endl;
            } else {
                printf("ipc_service: could not map the server's lane arena (%s); batches arrive in the pipe slots\n", why.c_str());
            }
        } else if (mirror_ != nullptr && mirror_->ext_version >= 2 && mirror_->arena_bytes[central_device] > 0 && opt_.direct_views) {
            hipIpcMemHandle_t h = *(hipIpcMemHandle_t*)&mirror_->arena[central_device];
            hipError_t e = hipIpcOpenMemHandle(&arena_, h, hipIpcMemLazyEnablePeerAccess);
            for (int attempt = 0; e != hipSuccess && attempt < 10; attempt++) {
                (void)hipGetLastError();
                usleep(25000);
                e = hipIpcOpenMemHandle(&arena_, h, hipIpcMemLazyEnablePeerAccess);
            }
            if (e == hipSuccess) {
                arena_bytes_ = mirror_->arena_bytes[central_device];
                mirror_->trainer_direct[central_device] = 1;
                __sync_synchronize();
                std::cout << "HIP: " << central_device << " lane arena opened (" << (arena_bytes_ >> 20) << " MiB): batches arrive as views\n";
            } else {
                (void)hipGetLastError();
                arena_ = nullptr;
                printf("ipc_service: could not open the server's lane arena ('%s'); batches arrive in the pipe slots\n", hipGetErrorString(e));
            }
        }
        semr_.resize(INTERBATCH_CON);
        semw_.resize(INTERBATCH_CON);
        const std::string sfx = opt_.suffix;
        for (int i = 0; i < INTERBATCH_CON; i++) {
            const std::string ssri = "sem_r_" + std::to_string(central_device) + "_" + std::to_string(i) + sfx;
            const std::string sswi = "sem_w_" + std::to_string(central_device) + "_" + std::to_string(i) + sfx;
            semr_[i] = sem_open(ssri.c_str(), O_CREAT | O_RDWR, 0666, 0);
            if (semr_[i] == SEM_FAILED) {
                printf("errno = %d\n", errno);
                return -1;
            }
            semw_[i] = sem_open(sswi.c_str(), O_CREAT | O_RDWR, 0666, 0);
            if (semw_[i] == SEM_FAILED) {
                printf("errno = %d\n", errno);
                return -1;
            }
            sem_post(semr_[i]);   // both pipe slots start free (ipc_cuda_kernel.cu:91)
        }
        current_pipe_ = 0;
        device_ = physical_device;
        slab_device_ = central_device;
        munmap(addr, sizeof(shmStruct));
        close(fd);
        return central_device;
    }

    void Wait()
    {
        sem_wait(semw_[current_pipe_]);
        if (mirror_ != nullptr && mirror_->server_state != 0) {      // the server woke every waiter before it stopped on an error
            printf("ipc_service: the sampling server stopped on an error; no batch was handed over\n");
            fflush(stdout);
            exit(EXIT_FAILURE);
        }
    }
    // the batch in the current pipe slot as byte offsets into the lane arena {ids, features, labels, agg_src, agg_dst}, or null
    const volatile int64_t* View() const
    {
        if (arena_ == nullptr || mirror_ == nullptr || mirror_->view_on[slab_device_][current_pipe_] == 0) return nullptr;
        return &mirror_->view[slab_device_][current_pipe_][0];
    }
    void* Arena() const { return arena_; }
    long long ArenaBytes() const { return arena_bytes_; }
    void Post()
    {
        sem_post(semr_[current_pipe_]);
        current_pipe_ = (current_pipe_ + 1) % INTERBATCH_CON;
    }
    int32_t* GetIds() { return (int32_t*)ids_[current_pipe_]; }
    float* GetFloatFeatures() { return (float*)float_features_[current_pipe_]; }
    int32_t* GetLabels() { return (int32_t*)labels_[current_pipe_]; }
    int32_t* GetAggSrc() { return (int32_t*)agg_src_[current_pipe_]; }
    int32_t* GetAggDst() { return (int32_t*)agg_dst_[current_pipe_]; }
    int32_t* GetNodeCounter() { return (int32_t*)node_counter_[current_pipe_]; }
    int32_t* GetEdgeCounter() { return (int32_t*)edge_counter_[current_pipe_]; }
    int32_t GetTrainStep() { return train_step_; }
    int32_t GetValidStep() { return valid_step_; }
    int32_t GetTestStep() { return test_step_; }
    int Device() const { return device_; }
    int CurrentPipe() const { return current_pipe_; }
    // host-visible counters of the batch in the current pipe slot, or null (then they are copied from the device)
    const volatile int32_t* CounterMirror() const { return mirror_ ? &mirror_->counters[slab_device_][current_pipe_][0] : nullptr; }

    // TB/ipc_cuda_kernel.cu:140-156 closes what it opened; so does this: every IPC handle, the lane arena (a chunked one unmapped
    // chunk by chunk and its address range given back; a plain one closed), the semaphores, the mirror object -- after which the
    // object is as new and Initialize() may attach to another server life.  (Round 4 kept the arena mapped "until the process
    // ends": a trainer that outlived its server kept the server's whole arena alive in HBM.)
    void Finalize()
    {
        std::vector<void*>* slots[MEMORY_USAGE] = {&ids_, &float_features_, &labels_, &agg_src_, &agg_dst_,
                                                   &node_counter_, &edge_counter_};
        (void)hipDeviceSynchronize();                      // nothing of this process still reads a view or a slot buffer
        for (int i = 0; i < (int)semw_.size(); i++) {
            for (auto* s : slots)
                if ((*s)[i] != nullptr && hipIpcCloseMemHandle((*s)[i]) != hipSuccess) (void)hipGetLastError();
            if (sem_close(semw_[i]) == -1) std::cout << "close sem " << i << " failed\n";
            sem_close(semr_[i]);
        }
        for (auto* s : slots) s->clear();
        semw_.clear();
        semr_.clear();
        if (arena_ != nullptr) {
            if (arena_chunks_ > 0) unmap_arena_chunks(arena_, arena_chunks_, arena_chunks_, arena_chunk_bytes_);
            else if (hipIpcCloseMemHandle(arena_) != hipSuccess) (void)hipGetLastError();
        }
        arena_ = nullptr;
        arena_bytes_ = 0;
        arena_chunks_ = 0;
        arena_chunk_bytes_ = 0;
        if (mirror_ != nullptr) munmap((void*)mirror_, sizeof(shmExt));
        mirror_ = nullptr;
        current_pipe_ = 0;
    }

private:
    std::vector<void*> ids_, float_features_, labels_, agg_src_, agg_dst_, node_counter_, edge_counter_;
    std::vector<sem_t*> semw_, semr_;
    int32_t train_step_ = 0, valid_step_ = 0, test_step_ = 0;
    int current_pipe_ = 0;
    int device_ = 0;        // physical device the tensors live on
    int slab_device_ = 0;   // index of this trainer's GPU in the server's slab / semaphore names
    volatile shmExt* mirror_ = nullptr;
    void* arena_ = nullptr;          // the server's lane arena, opened (direct views), or null
    long long arena_bytes_ = 0;
    int arena_chunks_ = 0;           // > 0: mapped from that many file descriptors of arena_chunk_bytes_ each
    long long arena_chunk_bytes_ = 0;
    TrainerOptions opt_;
};

static GPUIPCEnv* env = nullptr;
static int32_t h_node_counter[16];
static int32_t h_edge_counter[16];

// Whole-buffer tensors over the seven IPC buffers of each pipe slot, wrapped once (torch::from_blob is the expensive
// part of get_next: a TensorImpl + deleter context per call); a batch's tensors are as_strided views of them.
struct SlotTensors { torch::Tensor ids, feats, labels, src, dst; };
static SlotTensors slot_base[INTERBATCH_CON];
static bool slot_base_ready[INTERBATCH_CON] = {false, false};
static torch::Tensor arena_i32, arena_f32;     // the whole lane arena as int32 / float32 (direct views)
static bool arena_ready = false;
// extent (in elements) of the device allocation behind an IPC-opened pointer
static long long whole_buffer(void* p, size_t elem)
{
    hipDeviceptr_t base = nullptr;
    size_t bytes = 0;
    if (hipMemGetAddressRange(&base, &bytes, (hipDeviceptr_t)p) == hipSuccess && bytes >= elem) {
        const size_t off = (size_t)((char*)p - (char*)base);
        return (long long)((bytes - off) / elem);
    }
    (void)hipGetLastError();
    return 1ll << 36;   // unknown: the server's extent bounds every view (views never exceed the batch's counts)
}

void InitializeIPC()
{
    if (env != nullptr) {
        printf("ipc_service: initialize() while attached; call finalize() first\n");
        exit(EXIT_FAILURE);
    }
    env = new GPUIPCEnv();
    env->Initialize();
}

void FinalizeIPC()
{
    for (int i = 0; i < INTERBATCH_CON; i++) {
        slot_base[i] = SlotTensors();
        slot_base_ready[i] = false;
    }
    arena_i32 = torch::Tensor();
    arena_f32 = torch::Tensor();
    arena_ready = false;
    if (env == nullptr) return;
    env->Finalize();
    delete env;
    env = nullptr;
}

// 0: hipMemImportFromShareableHandle of this process's HIP runtime takes a pointer to the descriptor, 1: the descriptor's value,
// -1: neither worked (no chunked arenas: batches arrive in the pipe slots).  Exposed for the tests.
int VmmFdConvention() { return vmm_fd_convention(); }

// training_backend/ipc_cuda_kernel.cu:177-235 + training_backend/ipc_service.cpp:44-59
std::vector<torch::Tensor> get_next(int feature_dim)
{
    TORCH_CHECK(env != nullptr, "ipc_service.get_next(): not attached -- call initialize() first (and not after finalize())");
    env->Wait();
    if (const volatile int32_t* m = env->CounterMirror()) {
        for (int i = 0; i < 16; i++) {
            h_node_counter[i] = m[i];
            h_edge_counter[i] = m[16 + i];
        }
    } else {
        hipMemcpy(h_node_counter, env->GetNodeCounter(), 16 * sizeof(int32_t), hipMemcpyDeviceToHost);
        hipMemcpy(h_edge_counter, env->GetEdgeCounter(), 16 * sizeof(int32_t), hipMemcpyDeviceToHost);
        hipCheckError();
    }
    // (the counters hold INTRABATCH_CON * 3 + hop + 1 <= 16 words: six hops at most, whatever the word says)
    const int hop_num = std::min(std::max(h_node_counter[INTRABATCH_CON * 3 - 1], 0), 16 - INTRABATCH_CON * 3 - 1);
    const int pipe = env->CurrentPipe();
    if (const volatile int64_t* vw = env->View()) {
        // the batch is where the server's launch group left it: views of its lane inside the arena
        if (!arena_ready) {
            const auto dev = torch::Device(torch::kCUDA, env->Device());
            arena_i32 = torch::from_blob(env->Arena(), {env->ArenaBytes() / 4}, torch::TensorOptions().dtype(torch::kI32).device(dev));
            arena_f32 = torch::from_blob(env->Arena(), {env->ArenaBytes() / 4}, torch::TensorOptions().dtype(torch::kF32).device(dev));
            arena_ready = true;
        }
        const long long o_ids = vw[0] / 4, o_feat = vw[1] / 4, o_lab = vw[2] / 4, o_src = vw[3] / 4, o_dst = vw[4] / 4;
        std::vector<torch::Tensor> ret;
        ret.reserve(3 + 2 * hop_num);
        const long long n_total = std::max(h_node_counter[INTRABATCH_CON * 3 + hop_num], 0);
        ret.push_back(arena_i32.as_strided({n_total}, {1}, o_ids));
        ret.push_back(arena_f32.as_strided({n_total, (long long)feature_dim}, {(long long)feature_dim, 1}, o_feat));
        ret.push_back(arena_i32.as_strided({(long long)std::max(h_node_counter[INTRABATCH_CON * 3], 0)}, {1}, o_lab));
        for (int i = hop_num; i > 0; i--) {
            const long long n_edges = std::max(h_edge_counter[INTRABATCH_CON * 3 + i], 0);
            ret.push_back(arena_i32.as_strided({n_edges}, {1}, o_src));
            ret.push_back(arena_i32.as_strided({n_edges}, {1}, o_dst));
        }
        return ret;
    }
    SlotTensors& b = slot_base[pipe];
    if (!slot_base_ready[pipe]) {
        const auto dev = torch::Device(torch::kCUDA, env->Device());
        const auto i32 = torch::TensorOptions().dtype(torch::kI32).device(dev);
        const auto f32 = torch::TensorOptions().dtype(torch::kF32).device(dev);
        b.ids = torch::from_blob(env->GetIds(), {whole_buffer(env->GetIds(), 4)}, i32);
        b.feats = torch::from_blob(env->GetFloatFeatures(), {whole_buffer(env->GetFloatFeatures(), 4)}, f32);
        b.labels = torch::from_blob(env->GetLabels(), {whole_buffer(env->GetLabels(), 4)}, i32);
        b.src = torch::from_blob(env->GetAggSrc(), {whole_buffer(env->GetAggSrc(), 4)}, i32);
        b.dst = torch::from_blob(env->GetAggDst(), {whole_buffer(env->GetAggDst(), 4)}, i32);
        slot_base_ready[pipe] = true;
    }
    std::vector<torch::Tensor> ret;
    ret.reserve(3 + 2 * hop_num);
    const long long n_total = std::max(h_node_counter[INTRABATCH_CON * 3 + hop_num], 0);
    ret.push_back(b.ids.as_strided({n_total}, {1}));
    ret.push_back(b.feats.as_strided({n_total, (long long)feature_dim}, {(long long)feature_dim, 1}));
    ret.push_back(b.labels.as_strided({(long long)std::max(h_node_counter[INTRABATCH_CON * 3], 0)}, {1}));
    for (int i = hop_num; i > 0; i--) {   // cumulative edge prefixes, outermost block first
        const long long n_edges = std::max(h_edge_counter[INTRABATCH_CON * 3 + i], 0);
        ret.push_back(b.src.as_strided({n_edges}, {1}));
        ret.push_back(b.dst.as_strided({n_edges}, {1}));
    }
    return ret;
}

// training_backend/ipc_service.cpp:61-79
std::vector<int> get_block_size()
{
    std::vector<int> ret;
    const int hop_num = std::min(std::max(h_node_counter[INTRABATCH_CON * 3 - 1], 0), 16 - INTRABATCH_CON * 3 - 1);
    for (int i = hop_num; i > 0; i--) {
        ret.push_back(h_node_counter[INTRABATCH_CON * 3 + i]);
        ret.push_back(h_node_counter[INTRABATCH_CON * 3 + i - 1]);
    }
    return ret;
}

std::vector<int32_t> get_steps()
{
    TORCH_CHECK(env != nullptr, "ipc_service.get_steps(): not attached");
    return {env->GetTrainStep(), env->GetValidStep(), env->GetTestStep()};
}

void Synchronize()
{
    TORCH_CHECK(env != nullptr, "ipc_service.synchronize(): not attached");
    env->Post();
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m)
{
    m.def("get_next", &get_next, "dataset get next (HIP)");
    m.def("get_block_size", &get_block_size, "get dgl block size (HIP)");
    m.def("get_steps", &get_steps, "get steps (HIP)");
    m.def("initialize", &InitializeIPC, "InitializeIPC (HIP)");
    m.def("finalize", &FinalizeIPC, "FinalizeIPC (HIP)");
    m.def("synchronize", &Synchronize, "synchronize (HIP)");
    m.def("vmm_fd_convention", &VmmFdConvention, "how this process's HIP runtime takes a file-descriptor handle (this build only)");
}
