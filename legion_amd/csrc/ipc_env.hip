// ipc_env.hip -- server end of the server <-> trainer wire protocol.
//
// Reference: SS/engine/ipc_service.cu (CUDAIPCEnv :33-349), SS/engine/helper_multiprocess.cu
// (only sharedMemoryCreate/Open/Close are used).  The protocol is kept bit-for-bit:
//   * POSIX shm "simpleIPCshm" = struct { int32 steps[3]; IpcMemHandle(64 B) memHandle[8][2][7]; }, exactly that size
//     slot order: 0 ids, 1 features, 2 labels, 3 agg_src, 4 agg_dst, 5 node_counter, 6 edge_counter
//   * named semaphores sem_r_<dev>_<pipe> (trainer -> server, buffer free) and
//     sem_w_<dev>_<pipe> (server -> trainer, batch ready), created with value 0
//   * step schedule (train+valid)*epoch + test, valid/test batch = ceil(n_p / ceil(max n / 512))
// hipIpcMemHandle_t is 64 bytes like cudaIpcMemHandle_t.  LEGION_IPC_NAMESPACE (optional env)
// suffixes the shm/semaphore names so that independent servers (tests) can share a host.
#include "legion_core.h"
#include <sys/socket.h>
#include <sys/un.h>
#include <cerrno>
#include <cstddef>
#include <thread>

#include <fcntl.h>
#include <semaphore.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cerrno>
#include <cstddef>
#include <cstring>
#include <iostream>

static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size is part of the wire format");

// The slab is the reference's, byte for byte and no larger (SS/engine/ipc_service.cu:28-31): a trainer built from the
// reference opens it with shm_open + ftruncate(sizeof(its struct)) (TB/helper_multiprocess.cpp:30-37), which must not
// change its size.  This build's extra -- per (device, pipe slot) the 16 + 16 counters of the batch in that slot in
// HOST memory, so the trainer end needs no device-to-host copy per batch (the reference does two blocking 64-byte
// cudaMemcpy per get_next, TB/ipc_cuda_kernel.cu:186-187) -- lives in a shm object of its own, "legionIPCext<suffix>":
// registered with HIP, written by the GPU, opened by this build's ipc_service when it exists.
#define LEGION_SHM_EXT_MAGIC 0x4C47494F   /* "LGIO" */
typedef struct shmStruct_st {
    int32_t steps[3];
    hipIpcMemHandle_t memHandle[MAX_DEVICE][INTERBATCH_CON][MEMORY_USAGE];
} shmStruct;
static_assert(offsetof(shmStruct, memHandle) == 12 && sizeof(shmStruct) == 12 + MAX_DEVICE * INTERBATCH_CON * MEMORY_USAGE * 64,
              "the reference's slab layout is the wire format");
// Version 2 of the object (round 4) adds the "direct view" hand-over.  The server produces mini-batches in launch groups
// into per-lane buffers (server.hip) that all live in ONE device allocation per GPU, the lane arena; its IPC handle is
// published here.  A trainer end that can open it (this build's ipc_service) says so in trainer_direct[dev] BEFORE its first
// sem_post; from then on a hand-over is no GPU work at all: the server writes, per (device, pipe slot), where the five
// trainer-visible arrays of the batch start inside the arena and posts sem_w -- same semaphores, same two-slot order, and
// the reference-sized slab above stays valid for a trainer that knows nothing of this (it gets copies in the slot buffers).
extern "C" int64_t lg_scattered_info(void* ptr, int32_t* n_chunks);
extern "C" int lg_scattered_export_fd(void* ptr, int32_t index);
extern "C" int32_t lg_scattered_serve(void* ptr, const char* name);
typedef struct shmExt_st {
    int32_t ext_magic;                                         // LEGION_SHM_EXT_MAGIC once the mirror below is live
    int32_t ext_version;                                       // >= 2: the fields behind `counters` exist; 3: those behind `view` too
    int32_t server_state;                                      // 0 serving, 1 the server stopped on an error (trainers must not wait)
    int32_t ext_reserved;
    int32_t counters[MAX_DEVICE][INTERBATCH_CON][32];          // [0..15] node_counter, [16..31] edge_counter
    hipIpcMemHandle_t arena[MAX_DEVICE];                       // lane arena of each server GPU, valid when arena_bytes > 0
    int64_t arena_bytes[MAX_DEVICE];
    int32_t trainer_direct[MAX_DEVICE];                        // written by the trainer end: 1 = it opened the arena and takes views
    int32_t view_on[MAX_DEVICE][INTERBATCH_CON];               // 1: the batch in this slot is the view below, 0: it is in the slot's buffers
    int64_t view[MAX_DEVICE][INTERBATCH_CON][5];               // byte offsets into the arena: ids, features, labels, agg_src, agg_dst
    // version 3: the arena may be built from physical chunks mapped in shuffled order (storage.hip d_alloc_scattered: the gathers then
    // write a group's rows all over the HBM, +7 %), which hipIpcGetMemHandle cannot export.  arena_kind 1: `arena` is unused; the trainer
    // connects to the abstract unix socket "legion_arena_<arena_sock_pid>_<dev>", receives arena_chunks file descriptors (SCM_RIGHTS, in
    // mapping order), imports each (hipMemImportFromShareableHandle) and maps them back to back: arena_chunks x arena_chunk_bytes >= arena_bytes
    int32_t arena_kind[MAX_DEVICE];
    int32_t arena_chunks[MAX_DEVICE];
    int64_t arena_chunk_bytes[MAX_DEVICE];
    int32_t arena_sock_pid;
    int32_t ext_reserved2;
} shmExt;
#define LEGION_SHM_EXT_VERSION 3

typedef struct sharedMemoryInfo_st {
    void* addr;
    size_t size;
    int shmFd;
} sharedMemoryInfo;

static std::string ipc_suffix()
{
    const char* ns = getenv("LEGION_IPC_NAMESPACE");
    return ns ? std::string(ns) : std::string();
}

static int sharedMemoryCreate(const char* name, size_t sz, sharedMemoryInfo* info)
{
    info->size = sz;
    shm_unlink(name);   // the server owns the name and starts first: never inherit a crashed run's slab
    info->shmFd = shm_open(name, O_RDWR | O_CREAT | O_EXCL, 0777);
    if (info->shmFd < 0) return errno;
    if (ftruncate(info->shmFd, sz) != 0) return errno;
    info->addr = mmap(0, sz, PROT_READ | PROT_WRITE, MAP_SHARED, info->shmFd, 0);
    if (info->addr == MAP_FAILED || info->addr == NULL) return errno;
    return 0;
}

static void sharedMemoryClose(sharedMemoryInfo* info)
{
    if (info->addr) munmap(info->addr, info->size);
    if (info->shmFd >= 0) close(info->shmFd);
    info->addr = nullptr;
    info->shmFd = -1;
}

class HIPIPCEnv : public IPCEnv {
public:
    HIPIPCEnv(int32_t device_count, bool create_shm) : device_count_(device_count)
    {
        info_.addr = nullptr;
        info_.shmFd = -1;
        if (create_shm) {
            shm_name_ = std::string("simpleIPCshm") + ipc_suffix();
            if (sharedMemoryCreate(shm_name_.c_str(), sizeof(*shm_), &info_) != 0) {
                printf("Failed to create shared memory slab\n");
                exit(EXIT_FAILURE);
            }
            shm_ = (volatile shmStruct*)info_.addr;
            memset((void*)shm_, 0, sizeof(*shm_));
            // the counter mirror: a shm object of its own that the GPUs write straight into
            ext_info_.addr = nullptr;
            ext_info_.shmFd = -1;
            ext_name_ = std::string("legionIPCext") + ipc_suffix();
            if (lg::tuning().shm_mirror && sharedMemoryCreate(ext_name_.c_str(), sizeof(shmExt), &ext_info_) == 0) {
                ext_ = (volatile shmExt*)ext_info_.addr;
                memset((void*)ext_, 0, sizeof(shmExt));
                if (hipHostRegister(ext_info_.addr, sizeof(shmExt), hipHostRegisterPortable | hipHostRegisterMapped) == hipSuccess) {
                    void* dptr = nullptr;
                    if (hipHostGetDevicePointer(&dptr, ext_info_.addr, 0) == hipSuccess && dptr != nullptr) {
                        ext_dev_ = (shmExt*)dptr;
                        registered_ = true;
                    } else {
                        (void)hipHostUnregister(ext_info_.addr);
                    }
                }
                (void)hipGetLastError();
            } else {
                shm_unlink(ext_name_.c_str());      // no mirror: never leave an older run's object for a trainer to find
                ext_name_.clear();
            }
        } else {
            local_ = new shmStruct();
            memset(local_, 0, sizeof(*local_));
            shm_ = local_;
        }
        ids_.resize(device_count);
        float_features_.resize(device_count);
        labels_.resize(device_count);
        agg_src_.resize(device_count);
        agg_dst_.resize(device_count);
        node_counter_.resize(device_count);
        edge_counter_.resize(device_count);
        semr_.resize(device_count);
        semw_.resize(device_count);
    }

    // ipc_service.cu:60-128: the schedule every GPU follows.  Training: as many steps as the SMALLEST partition fills with raw batches (the
    // last, partial batch of every partition is dropped: (n - 1) / B).  Validation / testing: the LARGEST partition is cut into batches of
    // at most 512 seeds, and every partition spreads its seeds over that many steps (ceil division): all GPUs post the same number of
    // batches, of per-GPU sizes.  The three log lines and steps[0..2] of the slab are the contract with the trainer end.
    void Coordinate(BuildInfo* info) override
    {
        const int32_t gpus = info->partition_count;
        epoch_ = info->epoch;
        raw_batch_size_ = info->raw_batch_size;
        const int32_t eval_unit = 512;
        // (steps, per-GPU batch size) of an evaluation set: steps from its largest partition, sizes by ceil(n_i / steps)
        auto spread = [gpus](const std::vector<int32_t>& count, std::vector<int32_t>& per_gpu) {
            int32_t largest = 0;
            for (int32_t g = 0; g < gpus; g++) largest = std::max(largest, count[g]);
            const int32_t steps = (largest - 1) / eval_unit + 1;
            per_gpu.clear();
            for (int32_t g = 0; g < gpus; g++) per_gpu.push_back((count[g] - 1) / steps + 1);
            return steps;
        };
        int32_t smallest_train = 1000000000;
        for (int32_t g = 0; g < gpus; g++) smallest_train = std::min(smallest_train, info->training_set_num[g]);
        train_step_ = (smallest_train - 1) / raw_batch_size_;
        train_batch_size_.assign(gpus, raw_batch_size_);
        valid_step_ = spread(info->validation_set_num, valid_batch_size_);
        test_step_ = spread(info->testing_set_num, test_batch_size_);

        std::cout << "Train Steps: " << train_step_ << "\n";
        std::cout << "Valid Steps: " << valid_step_ << "\n";
        std::cout << "Test Steps: " << test_step_ << "\n";
        shm_->steps[0] = train_step_;
        shm_->steps[1] = valid_step_;
        shm_->steps[2] = test_step_;
    }

    int32_t GetMaxStep() override { return ((train_step_ + valid_step_) * epoch_) + test_step_; }

    // ipc_service.cu:134-195
    void InitializeSamplesBuffer(int32_t batch_size, int32_t num_ids, int32_t feature_dim, int32_t device_id,
                                 int32_t pipeline_depth) override
    {
        (void)feature_dim;
        SetGPUDevice(device_id);
        const std::string sfx = ipc_suffix();
        semr_[device_id].resize(pipeline_depth);
        semw_[device_id].resize(pipeline_depth);
        for (int32_t i = 0; i < pipeline_depth; i++) {
            auto slot = [&](int k) { return (void*)&shm_->memHandle[device_id][i][k]; };
            void* new_ids = lg_alloc_exported((int64_t)num_ids * sizeof(int32_t), slot(0), __FILE__, __LINE__);
            void* new_labels = lg_alloc_exported((int64_t)batch_size * sizeof(int32_t), slot(2), __FILE__, __LINE__);
            void* new_agg_src = lg_alloc_exported((int64_t)num_ids * sizeof(int32_t), slot(3), __FILE__, __LINE__);
            void* new_agg_dst = lg_alloc_exported((int64_t)num_ids * sizeof(int32_t), slot(4), __FILE__, __LINE__);
            void* new_node_counter = lg_alloc_exported(16 * sizeof(int32_t), slot(5), __FILE__, __LINE__);
            void* new_edge_counter = lg_alloc_exported(16 * sizeof(int32_t), slot(6), __FILE__, __LINE__);
            HIP_CALL(hipMemset(new_node_counter, 0, 64));
            HIP_CALL(hipMemset(new_edge_counter, 0, 64));
            ids_[device_id].push_back(new_ids);
            labels_[device_id].push_back(new_labels);
            agg_src_[device_id].push_back(new_agg_src);
            agg_dst_[device_id].push_back(new_agg_dst);
            node_counter_[device_id].push_back(new_node_counter);
            edge_counter_[device_id].push_back(new_edge_counter);

            const std::string ssri = "sem_r_" + std::to_string(device_id) + "_" + std::to_string(i) + sfx;
            const std::string sswi = "sem_w_" + std::to_string(device_id) + "_" + std::to_string(i) + sfx;
            // a crashed run (every HIP error path exits without Finalize) leaves semaphores with counts behind; a stale
            // sem_r count would let this server overwrite a buffer the trainer still reads, a stale sem_w count would
            // hand the trainer a batch that was never produced.  The server owns the names and starts first.
            sem_unlink(ssri.c_str());
            sem_unlink(sswi.c_str());
            semr_[device_id][i] = sem_open(ssri.c_str(), O_CREAT | O_EXCL | O_RDWR, 0666, 0);
            if (semr_[device_id][i] == SEM_FAILED) {
                printf("errno = %d\n", errno);
                return;
            }
            semw_[device_id][i] = sem_open(sswi.c_str(), O_CREAT | O_EXCL | O_RDWR, 0666, 0);
            if (semw_[device_id][i] == SEM_FAILED) {
                printf("errno = %d\n", errno);
                return;
            }
        }
        pipeline_depth_ = pipeline_depth;
    }

    // ipc_service.cu:197-207
    void InitializeFeaturesBuffer(int32_t batch_size, int32_t num_ids, int32_t feature_dim, int32_t device_id,
                                  int32_t pipeline_depth) override
    {
        (void)batch_size;
        SetGPUDevice(device_id);
        for (int32_t i = 0; i < pipeline_depth; i++) {
            void* new_features = lg_alloc_exported((int64_t)num_ids * feature_dim * sizeof(float),
                                                   (void*)&shm_->memHandle[device_id][i][1], __FILE__, __LINE__);
            float_features_[device_id].push_back(new_features);
        }
    }

    int32_t GetRawBatchsize() override { return raw_batch_size_; }

    // ipc_service.cu:213-228
    int32_t GetLocalBatchId(int32_t global_batch_id) override
    {
        int32_t local_batch_id = -1;
        if (global_batch_id < ((train_step_ + valid_step_) * epoch_)) {
            const int32_t epoch_batch_id = global_batch_id % (train_step_ + valid_step_);
            local_batch_id = epoch_batch_id < train_step_ ? epoch_batch_id : epoch_batch_id - train_step_;
        } else {
            local_batch_id = (global_batch_id - ((train_step_ + valid_step_) * epoch_)) % test_step_;
        }
        return local_batch_id;
    }

    // ipc_service.cu:230-238
    int32_t GetCurrentBatchsize(int32_t dev_id, int32_t current_mode) override
    {
        if (current_mode == TRAINMODE) return train_batch_size_[dev_id];
        if (current_mode == VALIDMODE) return valid_batch_size_[dev_id];
        return test_batch_size_[dev_id];
    }

    // ipc_service.cu:241-254
    int32_t GetCurrentMode(int32_t global_batch_id) override
    {
        if (global_batch_id < ((train_step_ + valid_step_) * epoch_)) {
            const int32_t epoch_batch_id = global_batch_id % (train_step_ + valid_step_);
            return epoch_batch_id < train_step_ ? TRAINMODE : VALIDMODE;
        }
        return TESTMODE;
    }

    int32_t* GetIds(int32_t d, int32_t p) override { return (int32_t*)ids_[d][p % pipeline_depth_]; }
    float* GetFloatFeatures(int32_t d, int32_t p) override { return (float*)float_features_[d][p % pipeline_depth_]; }
    int32_t* GetLabels(int32_t d, int32_t p) override { return (int32_t*)labels_[d][p % pipeline_depth_]; }
    int32_t* GetAggSrc(int32_t d, int32_t p) override { return (int32_t*)agg_src_[d][p % pipeline_depth_]; }
    int32_t* GetAggDst(int32_t d, int32_t p) override { return (int32_t*)agg_dst_[d][p % pipeline_depth_]; }
    int32_t* GetNodeCounter(int32_t d, int32_t p) override { return (int32_t*)node_counter_[d][p % pipeline_depth_]; }
    int32_t* GetEdgeCounter(int32_t d, int32_t p) override { return (int32_t*)edge_counter_[d][p % pipeline_depth_]; }

    // device address of the host-visible counter mirror of (device, pipe slot), or null when the slab could not be
    // registered (then the trainer end falls back to copying the counters from the device buffers)
    int32_t* GetCounterMirror(int32_t d, int32_t p) override
    {
        return ext_dev_ ? &ext_dev_->counters[d][p % pipeline_depth_][0] : nullptr;
    }
    void PublishMirror() override
    {
        if (!ext_dev_) return;
        ext_->ext_version = LEGION_SHM_EXT_VERSION;
        __sync_synchronize();
        ext_->ext_magic = LEGION_SHM_EXT_MAGIC;
    }
    // ---- direct-view hand-over (see shmExt) ----
    int32_t* HostCounterMirror(int32_t d, int32_t p) override
    {
        return ext_ ? (int32_t*)&ext_->counters[d][p % pipeline_depth_][0] : nullptr;
    }
    bool PublishArena(int32_t dev_id, void* base, int64_t bytes) override
    {
        if (!ext_ || !ext_dev_ || base == nullptr || bytes <= 0) return false;
        int32_t n_chunks = 0;
        const int64_t chunk_bytes = lg_scattered_info(base, &n_chunks);
        if (chunk_bytes > 0) {          // an arena of shuffled chunks: handed over as file descriptors (shmExt, version 3)
            char name[64];
            snprintf(name, sizeof(name), "legion_arena_%d_%d", (int)getpid(), dev_id);
            if (!lg_scattered_serve(base, name)) return false;      // (storage.hip: a thread hands the chunks' descriptors to a same-user peer)
            ext_->arena_kind[dev_id] = 1;
            ext_->arena_chunks[dev_id] = n_chunks;
            ext_->arena_chunk_bytes[dev_id] = chunk_bytes;
            ext_->arena_sock_pid = (int32_t)getpid();
            __sync_synchronize();
            ext_->arena_bytes[dev_id] = bytes;
            return true;
        }
        hipIpcMemHandle_t h;
        lg_ipc_export(&h, base, __FILE__, __LINE__);
        memcpy((void*)&ext_->arena[dev_id], &h, sizeof(h));
        __sync_synchronize();
        ext_->arena_bytes[dev_id] = bytes;
        return true;
    }
    bool TrainerTakesViews(int32_t dev_id) override { return ext_ && ext_->arena_bytes[dev_id] > 0 && ext_->trainer_direct[dev_id] == 1; }
    void SetView(int32_t dev_id, int32_t pipe, const int64_t* off5, const int32_t* counters32) override
    {
        if (!ext_) return;
        if (counters32) for (int i = 0; i < 32; i++) ext_->counters[dev_id][pipe][i] = counters32[i];
        if (off5) for (int i = 0; i < 5; i++) ext_->view[dev_id][pipe][i] = off5[i];
        ext_->view_on[dev_id][pipe] = off5 ? 1 : 0;
    }
    // The server stops on an error (corrupt batch): trainers blocked in sem_wait would wait for ever.  Say so where this
    // build's trainer end looks (server_state), wake every waiter, and take the names away.
    void AbortServing() override
    {
        if (ext_) ext_->server_state = 1;
        __sync_synchronize();
        const std::string sfx = ipc_suffix();
        for (int32_t i = 0; i < device_count_; i++)
            for (size_t j = 0; j < semw_[i].size(); j++) {
                if (semw_[i][j] != nullptr && semw_[i][j] != SEM_FAILED) { sem_post(semw_[i][j]); sem_post(semw_[i][j]); }
                sem_unlink(("sem_r_" + std::to_string(i) + "_" + std::to_string(j) + sfx).c_str());
                sem_unlink(("sem_w_" + std::to_string(i) + "_" + std::to_string(j) + sfx).c_str());
            }
        if (!shm_name_.empty()) shm_unlink(shm_name_.c_str());
        if (!ext_name_.empty()) shm_unlink(ext_name_.c_str());
    }
    bool IPCTryWait(int32_t dev_id, int32_t current_pipe) override { return sem_trywait(semr_[dev_id][current_pipe]) == 0; }

    void IPCPost(int32_t dev_id, int32_t current_pipe) override { sem_post(semw_[dev_id][current_pipe]); }
    void IPCWait(int32_t dev_id, int32_t current_pipe) override { sem_wait(semr_[dev_id][current_pipe]); }

    // ipc_service.cu:293-321
    void Finalize() override
    {
        const std::string sfx = ipc_suffix();
        for (int32_t i = 0; i < device_count_; i++) {
            if (ids_[i].empty()) continue;
            SetGPUDevice(i);
            for (int32_t j = 0; j < pipeline_depth_; j++) {
                d_free_space(ids_[i][j]);
                if (j < (int32_t)float_features_[i].size()) d_free_space(float_features_[i][j]);
                d_free_space(labels_[i][j]);
                d_free_space(agg_src_[i][j]);
                d_free_space(agg_dst_[i][j]);
                d_free_space(node_counter_[i][j]);
                d_free_space(edge_counter_[i][j]);
                if (sem_close(semw_[i][j]) == -1) std::cout << "close sem " << i << " " << j << " failed\n";
                sem_close(semr_[i][j]);
                const std::string ssri = "sem_r_" + std::to_string(i) + "_" + std::to_string(j) + sfx;
                const std::string sswi = "sem_w_" + std::to_string(i) + "_" + std::to_string(j) + sfx;
                sem_unlink(ssri.c_str());
                sem_unlink(sswi.c_str());
            }
            ids_[i].clear();
        }
        if (local_) {
            delete local_;
            local_ = nullptr;
        } else {
            if (registered_) (void)hipHostUnregister(ext_info_.addr);
            registered_ = false;
            ext_dev_ = nullptr;
            ext_ = nullptr;
            sharedMemoryClose(&info_);
            if (!shm_name_.empty()) shm_unlink(shm_name_.c_str());
            if (!ext_name_.empty()) {
                sharedMemoryClose(&ext_info_);
                shm_unlink(ext_name_.c_str());
            }
        }
        shm_ = nullptr;
    }

    int32_t GetTrainStep() override { return train_step_; }

private:
    volatile shmStruct* shm_ = nullptr;
    volatile shmExt* ext_ = nullptr;    // the counter mirror (its own shm object) ...
    shmExt* ext_dev_ = nullptr;         // ... as the GPUs see it (registered host memory)
    bool registered_ = false;
    shmStruct* local_ = nullptr;
    sharedMemoryInfo info_, ext_info_ = {nullptr, 0, -1};
    std::string shm_name_, ext_name_;
    std::vector<std::vector<void*>> ids_, float_features_, labels_, agg_src_, agg_dst_, node_counter_, edge_counter_;
    std::vector<std::vector<sem_t*>> semr_, semw_;
    int32_t raw_batch_size_ = 0;
    std::vector<int32_t> train_batch_size_, valid_batch_size_, test_batch_size_;
    int32_t device_count_ = 0;
    int32_t train_step_ = 0, valid_step_ = 0, test_step_ = 0;
    int32_t epoch_ = 0;
    int32_t pipeline_depth_ = INTERBATCH_CON;
};

IPCEnv* NewIPCEnvImpl(int32_t device_count, bool create_shm) { return new HIPIPCEnv(device_count, create_shm); }

// ---- C API (ipc_service.h:35 NewIPCEnv + the host-only step arithmetic) ---------------------
extern "C" LegionIPCEnv* NewIPCEnv(int32_t device_count)
{
    // LEGION_IPC_LOCAL=1: keep the slab in process memory (step arithmetic only, no /dev/shm entry)
    const char* local = getenv("LEGION_IPC_LOCAL");
    return reinterpret_cast<LegionIPCEnv*>(NewIPCEnvImpl(device_count, !(local && local[0] == '1')));
}

extern "C" void legion_ipc_coordinate(LegionIPCEnv* e, int32_t partition_count, const int32_t* train_num,
                                      const int32_t* valid_num, const int32_t* test_num, int32_t raw_batch_size,
                                      int32_t epoch)
{
    if (!e) { printf("invalid ipc env ptr\n"); return; }
    BuildInfo info;
    info.partition_count = partition_count;
    info.training_set_num.assign(train_num, train_num + partition_count);
    info.validation_set_num.assign(valid_num, valid_num + partition_count);
    info.testing_set_num.assign(test_num, test_num + partition_count);
    info.raw_batch_size = raw_batch_size;
    info.epoch = epoch;
    reinterpret_cast<IPCEnv*>(e)->Coordinate(&info);
}

extern "C" int32_t legion_ipc_train_step(LegionIPCEnv* e) { return reinterpret_cast<IPCEnv*>(e)->GetTrainStep(); }
extern "C" int32_t legion_ipc_max_step(LegionIPCEnv* e) { return reinterpret_cast<IPCEnv*>(e)->GetMaxStep(); }
extern "C" int32_t legion_ipc_current_mode(LegionIPCEnv* e, int32_t g) { return reinterpret_cast<IPCEnv*>(e)->GetCurrentMode(g); }
extern "C" int32_t legion_ipc_local_batch_id(LegionIPCEnv* e, int32_t g) { return reinterpret_cast<IPCEnv*>(e)->GetLocalBatchId(g); }
extern "C" int32_t legion_ipc_current_batchsize(LegionIPCEnv* e, int32_t d, int32_t m)
{
    return reinterpret_cast<IPCEnv*>(e)->GetCurrentBatchsize(d, m);
}
extern "C" void legion_ipc_finalize(LegionIPCEnv* e)
{
    if (!e) return;
    IPCEnv* env = reinterpret_cast<IPCEnv*>(e);
    env->Finalize();
    delete env;
}
