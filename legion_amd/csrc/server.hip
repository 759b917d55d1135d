// server.hip -- StorageManagement, GPURunner, GPUServer and the binary / pybind-style entry.
//
// Reference: SS/storage/storage_management.cu (EnableP2PAccess :5-23, ReadMetaFIle :29-98,
// LoadGraph :100-115, LoadFeature :118-232, Initialze :234-269), SS/engine/server.cu (GPUServer
// :44-169, GPURunner :170-365, PreSCLoop/RunnerLoop :29-42), SS/main.cu,
// sampling_server/sampling_server.cpp:7.
//
// Kept: the meta_config -> argv -> op DAG chain, one host thread per GPU, the op order
// (3*(hops+1)+1 ops), the two-deep pipe with semaphores, the log lines ("System is ready for
// serving" is the trainer's start signal).  Changed: all ops of a batch are enqueued on one HIP
// stream without any host synchronisation in between (the kernels read their sizes on the
// device), and the batch is posted only after the LAST op has completed (the reference records
// only stream 0's last event).  Full CSR / feature table go to HBM when they fit
// (288 GB per MI355X), to mapped pinned host memory otherwise.
#include "legion_core.h"

#include <pthread.h>
extern "C" void* d_alloc_scattered_exportable(int64_t num_bytes, int32_t chunk_mb);
#include "runner_schedule.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <atomic>
#include <cstring>
#include <deque>
#include <map>
#include <fstream>
#include <iostream>
#include <sstream>
#include <thread>

// ---------------------------------------------------------------------------------------------
// Reads exactly `bytes` bytes.  A file that exists but is SHORTER than meta_config says is a corrupt or mismatched
// data set: the reference mmaps it and reads past the end (storage_management_impl.cuh:46-159); here the server
// says what is wrong and stops instead of sampling from zero-filled tables.
static void read_file_into(const std::string& path, void* dst, size_t bytes)
{
    int fd = open(path.c_str(), O_RDONLY);
    if (fd == -1) {
        std::cout << "cannout open file: " << path << "\n";
        exit(EXIT_FAILURE);
    }
    size_t done = 0;
    char* p = (char*)dst;
    while (done < bytes) {
        ssize_t r = read(fd, p + done, bytes - done);
        if (r <= 0) break;
        done += (size_t)r;
    }
    close(fd);
    if (done < bytes) {
        std::cout << "file too short: " << path << " holds " << done << " bytes, meta_config implies " << bytes << "\n" << std::flush;
        exit(EXIT_FAILURE);
    }
}

static int64_t file_size(const std::string& path)
{
    struct stat st;
    if (stat(path.c_str(), &st) != 0) return -1;
    return (int64_t)st.st_size;
}

// a host staging buffer pushed to HBM, or mapped pinned memory if HBM placement is not possible
struct Placement {
    void* dev_ptr = nullptr;
    bool in_hbm = false;
};

// HBM budget of the full tables (CSR + features): 70 % of what was free when loading started, shared by ALL of them --
// three tables that each fit must not over-commit together; the rest stays for caches, pools and the trainer.
static int64_t g_hbm_budget = -1;

static Placement place_table(const std::string& path, size_t bytes, bool want_hbm, bool required)
{
    Placement pl;
    void* pinned = nullptr;
    HIP_CALL(hipHostMalloc(&pinned, bytes ? bytes : 16, hipHostMallocMapped));
    const bool present = file_size(path) >= 0;
    if (present) {
        read_file_into(path, pinned, bytes);
    } else if (required) {
        std::cout << "cannout open file: " << path << "\n" << std::flush;
        exit(EXIT_FAILURE);
    } else {
        memset(pinned, 0, bytes);       // optional table absent (v2 of the reference reads neither features nor labels)
    }
    if (want_hbm) {
        if (g_hbm_budget < 0) {
            size_t free_b = 0, total_b = 0;
            HIP_CALL(hipMemGetInfo(&free_b, &total_b));
            g_hbm_budget = (int64_t)(free_b / 10 * 7);
        }
        if ((int64_t)bytes <= g_hbm_budget) {
            g_hbm_budget -= (int64_t)bytes;
            void* d = d_alloc_space((int64_t)bytes);
            HIP_CALL(hipMemcpy(d, pinned, bytes, hipMemcpyHostToDevice));
            HIP_CALL(hipHostFree(pinned));
            pl.dev_ptr = d;
            pl.in_hbm = true;
            return pl;
        }
    }
    HIP_CALL(hipHostGetDevicePointer(&pl.dev_ptr, pinned, 0));
    return pl;
}

class StorageManagement {
public:
    // SS/storage/storage_management.cu:5-23
    void EnableP2PAccess()
    {
        int32_t device_count = legion_device_count();
        for (int32_t i = 0; i < device_count; i++) {
            HIP_CALL(hipSetDevice(i));
            for (int32_t j = 0; j < device_count; j++) {
                if (j == i) continue;
                int32_t accessible = 0;
                HIP_CALL(hipDeviceCanAccessPeer(&accessible, i, j));
                if (accessible) {
                    hipError_t e = hipDeviceEnablePeerAccess(j, 0);
                    if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) {
                        printf("HIP failure %s:%d: '%s'\n", __FILE__, __LINE__, hipGetErrorString(e));
                        exit(EXIT_FAILURE);
                    }
                    (void)hipGetLastError();
                }
            }
        }
    }

    // SS/storage/storage_management.cu:29-98: ten fields in memory mode; disk mode (in_memory_mode == 0) appends partition, the
    // two SSD parameters (read and logged; the SSD reader is unreleased upstream) and the two capacities of the hybrid
    // CPU-cache / GPU-cache tier (:85-94)
    bool ReadMetaFIle(BuildInfo* info, int32_t in_memory_mode)
    {
        std::ifstream Metafile("./meta_config");
        if (!Metafile.is_open()) {
            std::cout << "unable to open meta config file" << "\n";
            return false;
        }
        std::string buff;
        getline(Metafile, buff);
        std::istringstream iss(buff);
        iss >> dataset_path_;
        std::cout << "Dataset path:       " << dataset_path_ << "\n";
        iss >> raw_batch_size_;
        std::cout << "Raw Batchsize:      " << raw_batch_size_ << "\n";
        info->raw_batch_size = raw_batch_size_;
        iss >> node_num_;
        std::cout << "Graph nodes num:    " << node_num_ << "\n";
        iss >> edge_num_;
        std::cout << "Graph edges num:    " << edge_num_ << "\n";
        iss >> float_feature_len_;
        std::cout << "Feature dim:        " << float_feature_len_ << "\n";
        iss >> training_set_num_;
        std::cout << "Training set num:   " << training_set_num_ << "\n";
        iss >> validation_set_num_;
        std::cout << "Validation set num: " << validation_set_num_ << "\n";
        iss >> testing_set_num_;
        std::cout << "Testing set num:    " << testing_set_num_ << "\n";
        iss >> cache_memory_;
        std::cout << "Cache memory:       " << cache_memory_ << "\n";
        iss >> epoch_;
        std::cout << "Train epoch:        " << epoch_ << "\n";
        info->epoch = epoch_;
        if (!in_memory_mode) {
            iss >> partition_;
            std::cout << "Partition?:         " << partition_ << "\n";
            iss >> num_ssd_;
            std::cout << "SSD Num?:           " << num_ssd_ << "\n";
            iss >> num_queues_per_ssd_;
            std::cout << "Q/SSD    ?:         " << num_queues_per_ssd_ << "\n";
            iss >> cpu_cache_capacity_;
            std::cout << "CPU Cache Capacity: " << cpu_cache_capacity_ << "\n";
            iss >> gpu_cache_capacity_;
            std::cout << "GPU Cache Capacity: " << gpu_cache_capacity_ << "\n";
            if (iss.fail() || cpu_cache_capacity_ < 0 || gpu_cache_capacity_ < 0) {
                std::cout << "meta_config: disk mode needs fifteen fields (... partition ssd_num queues_per_ssd cpu_cache_capacity "
                             "gpu_cache_capacity)\n" << std::flush;
                return false;
            }
        }
        return true;
    }

    // SS/storage/storage_management.cu:100-115; file formats SURVEY.md A.5
    void LoadGraph(BuildInfo* info)
    {
        info->total_edge_num = edge_num_;
        const bool hbm = want_hbm();
        Placement ip = place_table(dataset_path_ + "edge_src", (size_t)(node_num_ + 1) * sizeof(int64_t), hbm, true);
        Placement ci = place_table(dataset_path_ + "edge_dst", (size_t)edge_num_ * sizeof(int32_t), hbm, true);
        {   // the two files must describe one graph: indptr[N] is the edge count meta_config gives
            int64_t last = 0;
            HIP_CALL(hipMemcpy(&last, (int64_t*)ip.dev_ptr + node_num_, sizeof(int64_t), hipMemcpyDefault));
            if (last != edge_num_) {
                std::cout << "data set mismatch: edge_src[N] = " << last << " but meta_config says " << edge_num_ << " edges\n" << std::flush;
                exit(EXIT_FAILURE);
            }
        }
        info->csr_node_index = (int64_t*)ip.dev_ptr;
        info->csr_dst_node_ids = (int32_t*)ci.dev_ptr;
        std::cout << "Topology placement: " << (ip.in_hbm && ci.in_hbm ? "HBM" : "pinned host") << "\n";
    }

    // SS/storage/storage_management.cu:118-232
    void LoadFeature(BuildInfo* info)
    {
        const int32_t partition_count = info->partition_count;
        info->training_set_ids.assign(partition_count, {});
        info->training_labels.assign(partition_count, {});
        info->validation_set_ids.assign(partition_count, {});
        info->validation_labels.assign(partition_count, {});
        info->testing_set_ids.assign(partition_count, {});
        info->testing_labels.assign(partition_count, {});
        std::vector<int32_t> training_ids(training_set_num_), validation_ids(validation_set_num_),
            testing_ids(testing_set_num_), all_labels(node_num_, 0), partition_index;
        read_file_into(dataset_path_ + "trainingset", training_ids.data(), training_ids.size() * 4);
        read_file_into(dataset_path_ + "validationset", validation_ids.data(), validation_ids.size() * 4);
        read_file_into(dataset_path_ + "testingset", testing_ids.data(), testing_ids.size() * 4);
        // v2 of the reference leaves features uninitialised and labels zero (:162,:164); this build
        // reads both files when they exist and zero-fills otherwise.
        Placement fp = place_table(dataset_path_ + "features", (size_t)node_num_ * float_feature_len_ * sizeof(float),
                                   want_hbm(), false);
        if (file_size(dataset_path_ + "labels") >= 0)
            read_file_into(dataset_path_ + "labels", all_labels.data(), all_labels.size() * 4);
        const bool have_partition = file_size(dataset_path_ + "partition") >= (int64_t)node_num_ * 4;
        if (have_partition) {
            partition_index.resize(node_num_);
            read_file_into(dataset_path_ + "partition", partition_index.data(), partition_index.size() * 4);
        } else {
            std::cout << "cannout open file: " << dataset_path_ + "partition" << "\n";
        }
        std::cout << "Finish Reading All Files\n";
        for (const std::vector<int32_t>* set : {&training_ids, &validation_ids, &testing_ids})
            for (int32_t tid : *set)
                if (tid < 0 || tid >= node_num_) {     // would index every table out of bounds on the device
                    std::cout << "data set mismatch: seed id " << tid << " outside [0, " << node_num_ << ")\n" << std::flush;
                    exit(EXIT_FAILURE);
                }
        int trainingset_count = 0;
        for (int32_t tid : training_ids) {                                  // :171-184
            const int32_t part_id = have_partition ? partition_index[tid] : tid % partition_count;
            if (part_id < partition_count) {
                info->training_set_ids[part_id].push_back(tid);
                trainingset_count++;
            }
        }
        std::cout << "training set count " << trainingset_count << "\n";
        for (int32_t tid : validation_ids) info->validation_set_ids[tid % partition_count].push_back(tid);
        for (int32_t tid : testing_ids) info->testing_set_ids[tid % partition_count].push_back(tid);
        for (int32_t p = 0; p < partition_count; p++) {
            for (int32_t id : info->training_set_ids[p]) info->training_labels[p].push_back(all_labels[id]);
            for (int32_t id : info->validation_set_ids[p]) info->validation_labels[p].push_back(all_labels[id]);
            for (int32_t id : info->testing_set_ids[p]) info->testing_labels[p].push_back(all_labels[id]);
            info->training_set_num.push_back((int32_t)info->training_set_ids[p].size());
            info->validation_set_num.push_back((int32_t)info->validation_set_ids[p].size());
            info->testing_set_num.push_back((int32_t)info->testing_set_ids[p].size());
        }
        info->host_float_feature = (float*)fp.dev_ptr;
        info->float_feature_len = float_feature_len_;
        info->total_num_nodes = node_num_;
        std::cout << "Feature placement:  " << (fp.in_hbm ? "HBM" : "pinned host") << "\n";
    }

    // SS/storage/storage_management.cu:234-269
    bool Initialze(int32_t partition_count, int32_t in_memory_mode)
    {
        BuildInfo* info = new BuildInfo();
        EnableP2PAccess();
        info->partition_count = partition_count;
        if (!ReadMetaFIle(info, in_memory_mode)) return false;
        SetGPUDevice(0);
        LoadGraph(info);
        LoadFeature(info);
        env_ = NewIPCEnvImpl(partition_count, true);
        env_->Coordinate(info);
        feature_ = NewCompleteFeatureStorage();
        feature_->Build(info, in_memory_mode);
        graph_ = NewCompleteGraphStorage();
        graph_->Build(info);
        hipCheckError();
        cache_ = new UnifiedCache();
        const int32_t train_step = env_->GetTrainStep();
        SetGPUDevice(0);
        cache_->Initialize(cache_memory_, float_feature_len_, train_step, partition_count, cpu_cache_capacity_, gpu_cache_capacity_);   // :266
        std::cout << "Storage Initialized\n";
        info_ = info;
        return true;
    }

    GraphStorage* GetGraph() { return graph_; }
    FeatureStorage* GetFeature() { return feature_; }
    UnifiedCache* GetCache() { return cache_; }
    IPCEnv* GetIPCEnv() { return env_; }

private:
    static bool want_hbm() { return lg::tuning().table_placement == 0; }   // LegionTuning: "hbm" (default) | "pinned"
    std::string dataset_path_;
    int32_t raw_batch_size_ = 0, node_num_ = 0, float_feature_len_ = 0;
    int64_t edge_num_ = 0, cache_memory_ = 0;
    int32_t training_set_num_ = 0, validation_set_num_ = 0, testing_set_num_ = 0, epoch_ = 0;
    int32_t partition_ = 0, num_ssd_ = 0, num_queues_per_ssd_ = 0, cpu_cache_capacity_ = 0, gpu_cache_capacity_ = 0;   // disk mode
    GraphStorage* graph_ = nullptr;
    FeatureStorage* feature_ = nullptr;
    UnifiedCache* cache_ = nullptr;
    IPCEnv* env_ = nullptr;
    BuildInfo* info_ = nullptr;
};

// =============================================================================================
struct LegionPipeline;
extern "C" LegionPipeline* legion_pipeline_create(LegionGraphStorage* graph, LegionFeatureStorage* feature,
                                                  LegionUnifiedCache* cache, int32_t dev_id, int32_t batch_size,
                                                  const int32_t* fanout, int32_t hop_num, int32_t group_size,
                                                  int32_t slots, int64_t feature_rows, int32_t use_graph);
extern "C" int32_t legion_pipeline_submit_ex(LegionPipeline* p, int32_t counter0, int32_t mode, int32_t n_active, int32_t batch_size);
extern "C" void legion_pipeline_prepare(LegionPipeline* p, int32_t mode, int32_t n_active, int32_t batch_size);
extern "C" legion_stream_t legion_pipeline_stream(LegionPipeline* p);
extern "C" void* legion_pipeline_slot_done_event(LegionPipeline* p, int32_t slot);
extern "C" LegionMemoryPool* legion_pipeline_pool(LegionPipeline* p, int32_t slot, int32_t lane);
extern "C" void legion_pipeline_destroy(LegionPipeline* p);
extern "C" void legion_pipeline_wait(LegionPipeline* p, int32_t slot);
extern "C" void legion_pipeline_wait_sleeping(LegionPipeline* p, int32_t slot, int32_t spin_us);
extern "C" void legion_pipeline_set_gathers(LegionPipeline* p, int32_t on);
extern "C" int32_t legion_pipeline_bulk_enable_shared(LegionPipeline* p, const PoolArena* arena);
extern "C" int32_t legion_pipeline_submit_bulk_inproc(LegionPipeline* p, int32_t counter0, int32_t mode, int32_t n_active, int32_t batch_size);

class GPURunner : public Runner {
public:
    // SS/engine/server.cu:172-273
    void Initialize(RunnerParams* params) override
    {
        SetGPUDevice(params->device_id);
        local_dev_id_ = params->device_id;
        UnifiedCache* cache = (UnifiedCache*)(params->cache);
        FeatureStorage* feature = (FeatureStorage*)(params->feature);
        IPCEnv* env = (IPCEnv*)(params->env);
        env_ = env;

        streams_.resize(INTRABATCH_CON);
        for (int i = 0; i < INTRABATCH_CON; i++) HIP_CALL(hipStreamCreate(&streams_[i]));

        // The reference sizes every buffer from the raw (training) batch size (server.cu:181-199) although
        // validation/test batches can be larger (ceil(n / ceil(max n / 512)), ipc_service.cu:91-115) and
        // then overrun `labels`; here the pool is sized for the largest batch of any mode.
        int batch_size = env->GetRawBatchsize();
        for (int mode = TRAINMODE; mode <= TESTMODE; mode++)
            batch_size = std::max(batch_size, (int)env->GetCurrentBatchsize(local_dev_id_, mode));
        const int hop_num = (int)params->fanout.size();
        std::vector<int32_t> fanout(params->fanout.begin(), params->fanout.end());

        op_num_ = (hop_num + 1) * INTRABATCH_CON + 1;
        op_factory_.resize(op_num_);
        op_factory_[0] = NewBatchGenerateOP(0);
        op_factory_[1] = NewCacheLookupOP(1);
        op_factory_[2] = NewSSDIOSubmitOP(2);
        for (int i = 0; i < hop_num; i++) {
            op_factory_[INTRABATCH_CON * i + 3] = NewRandomSampleOP(INTRABATCH_CON * i + 3);
            op_factory_[INTRABATCH_CON * i + 4] = NewCacheLookupOP(INTRABATCH_CON * i + 4);
            op_factory_[INTRABATCH_CON * i + 5] = NewSSDIOSubmitOP(INTRABATCH_CON * i + 5);
        }
        op_factory_[op_num_ - 1] = NewSSDIOCompleteOP(op_num_ - 1);

        interbatch_concurrency_ = INTERBATCH_CON;
        const int total_num_nodes = feature->TotalNodeNum();
        cache->InitializeCacheController(local_dev_id_, total_num_nodes);

        memorypool_ = new MemoryPool(interbatch_concurrency_);
        float_feature_len_ = feature->GetFloatFeatureLen();
        lg_pool_alloc_private(memorypool_, local_dev_id_, total_num_nodes, batch_size, fanout.data(), hop_num,
                              float_feature_len_);
        num_ids_ = memorypool_->num_ids;
        env->InitializeSamplesBuffer(batch_size, num_ids_, float_feature_len_, local_dev_id_, interbatch_concurrency_);
        current_pipe_ = 0;
        for (int i = 0; i < INTERBATCH_CON; i++) {
            memorypool_->SetSampledIds(env->GetIds(local_dev_id_, i), i);
            memorypool_->SetLabels(env->GetLabels(local_dev_id_, i), i);
            memorypool_->SetAggSrcOf(env->GetAggSrc(local_dev_id_, i), i);
            memorypool_->SetAggDstOf(env->GetAggDst(local_dev_id_, i), i);
            memorypool_->SetNodeCounter(env->GetNodeCounter(local_dev_id_, i), i);
            memorypool_->SetEdgeCounter(env->GetEdgeCounter(local_dev_id_, i), i);
        }
        // PreSC runs before the feature buffers exist; BatchGenerate/RandomSample never touch them.

        events_.resize(op_num_);
        op_params_.resize(op_num_);
        for (int i = 0; i < op_num_; i++) {
            op_params_[i] = new OpParams();
            op_params_[i]->device_id = local_dev_id_;
            op_params_[i]->stream = streams_[0];   // one in-order stream per batch, see file header
            HIP_CALL(hipEventCreateWithFlags(&events_[i], hipEventDisableTiming));
            op_params_[i]->event = events_[i];
            op_params_[i]->memorypool = memorypool_;
            op_params_[i]->cache = cache;
            op_params_[i]->graph = params->graph;
            op_params_[i]->feature = feature;
            op_params_[i]->env = env;
            op_params_[i]->in_memory = params->in_memory;
            op_params_[i]->hop_num = hop_num;
            op_params_[i]->neighbor_count = 0;
        }
        for (int i = 0; i < hop_num; i++)
            op_params_[INTRABATCH_CON * i + INTRABATCH_CON]->neighbor_count = params->fanout[i];
    }

    // SS/engine/server.cu:275-283
    void InitializeFeaturesBuffer(RunnerParams* params) override
    {
        SetGPUDevice(local_dev_id_);
        HIP_CALL(hipDeviceSynchronize());
        ReportErrors(memorypool_);          // a PreSC epoch over corrupt batches would size the caches from garbage: stop here
        UnifiedCache* cache = (UnifiedCache*)(params->cache);
        int32_t num_ids = int32_t((cache->MaxIdNum(local_dev_id_)) * 1.2);
        // MaxIdNum was measured on training batches; validation/test batches can be larger than the
        // raw batch (see Initialize): scale the reference's 1.2x rule by that ratio, never beyond num_ids
        IPCEnv* env_ = (IPCEnv*)(params->env);
        int64_t scaled = (int64_t)num_ids * ((memorypool_->batch_size + env_->GetRawBatchsize() - 1) / env_->GetRawBatchsize());
        if (scaled > memorypool_->num_ids) scaled = memorypool_->num_ids;
        num_ids = (int32_t)scaled;
        if (num_ids < 1) num_ids = 1;
        IPCEnv* env = (IPCEnv*)(params->env);
        // The 1.2 x rule sizes the LANES (hundreds of them).  The two pipe-slot buffers -- what a trainer end that does not take views
        // reads its rows from, and what the operator-by-operator Runner gathers into -- take the worst case of a batch, num_ids rows,
        // whenever the pair costs less than a tenth of the free HBM: no batch is ever truncated there (the reference sizes them by the
        // rule and overruns, server.cu:277).  LegionTuning.runner_overflow = 0 keeps the rule for them too.
        lane_rule_rows_ = num_ids;
        int64_t slot_rows = num_ids;
        {
            size_t free_b = 0, total_b = 0;
            HIP_CALL(hipMemGetInfo(&free_b, &total_b));
            const int64_t worst_bytes = (int64_t)memorypool_->num_ids * float_feature_len_ * (int64_t)sizeof(float);
            if (lg::tuning().runner_overflow != 0 && (int64_t)interbatch_concurrency_ * worst_bytes <= (int64_t)(free_b / 10)) slot_rows = memorypool_->num_ids;
        }
        env->InitializeFeaturesBuffer(0, (int32_t)slot_rows, float_feature_len_, local_dev_id_, interbatch_concurrency_);
        for (int i = 0; i < interbatch_concurrency_; i++)
            memorypool_->SetFloatFeatures(env->GetFloatFeatures(local_dev_id_, i), i);
        memorypool_->feature_rows = slot_rows;
        memorypool_->grid_rows_hint = num_ids;      // launches into the pipe slots are sized for the usual batch, not for the buffers' worst case
        // from here on every batch's counters are also written to the slab's host-visible mirror (lane-group path)
        if (use_groups_ && env->GetCounterMirror(local_dev_id_, 0) != nullptr) env->PublishMirror();
    }

    // SS/engine/server.cu:285-300
    void RunPreSc(RunnerParams* params) override
    {
        SetGPUDevice(local_dev_id_);
        memorypool_->SetCurrentMode(0);
        memorypool_->SetIter(params->global_batch_id);
        for (int i = 0; i < op_num_; i += INTRABATCH_CON) {
            op_params_[i]->is_presc = true;
            op_factory_[i]->run(op_params_[i]);
        }
        HIP_CALL(hipEventSynchronize(op_params_[op_num_ - 1]->event));
    }

    // SS/engine/server.cu:302-332.  The reference produces ONE mini-batch per call, ~16 launches and three blocking
    // read-backs each.  Here a call still hands ONE batch over -- in order, through the same two semaphores per pipe slot --
    // but the batches are PRODUCED in launch groups (pipeline.hip: every kernel runs with grid.y = lanes, a group is one
    // hipGraph replay per stream), up to three groups in flight, into per-lane buffers that all live in ONE exported device
    // allocation, the lane arena.  How a finished batch reaches the trainer end (LegionTuning.runner_handover):
    //   views   the trainer end opened the arena (this build's ipc_service says so before its first sem_post): the group ran
    //           the whole path, full-width gathers included -- the arrangement bench.py times -- and the hand-over is a few
    //           host stores (where the batch's five arrays start inside the arena, its counters) and sem_post.  No GPU work
    //           per batch at all.
    //   gather  any other trainer end (it opens only the reference's slab): the groups run the sampler phase; per batch ONE
    //           launch gathers the lane's rows straight into the pipe slot's feature buffer and copies its ids / edges /
    //           labels / counters there (deliver_slice in gather_kernel).  Bound by the launch -> completion latency of one
    //           small kernel with two slots in flight.
    //   (a third kind -- whole groups + one pure copy launch per batch into the pipe slot -- was built in round 4, measured slower than
    //   `gather` at every batch size, the rows crossing HBM twice, and removed in round 5: DESIGN_HISTORY.md)
    // A slot's lanes are reused by a later group only when the trainer has RELEASED every batch of the group that used
    // them (a view is read in place; a copied / gathered batch was posted before it was released): the semaphore token that
    // lets batch k through says batch k-2 was released.  LEGION_RUNNER_GRAPH=0: the operator-by-operator path of the
    // reference (one batch per call, eager launches).
    void RunOnce(RunnerParams* params) override
    {
        SetGPUDevice(local_dev_id_);
        IPCEnv* env = (IPCEnv*)(params->env);
        const int32_t batch_id = params->global_batch_id;
        if (!use_groups_) {
            mode_ = env->GetCurrentMode(batch_id);
            memorypool_->SetCurrentMode(mode_);
            memorypool_->SetIter(env->GetLocalBatchId(batch_id));
            env->IPCWait(local_dev_id_, current_pipe_);
            for (int i = 0; i < op_num_; i++) {
                op_params_[i]->is_presc = false;
                op_factory_[i]->run(op_params_[i]);
            }
            HIP_CALL(hipEventSynchronize(op_params_[op_num_ - 1]->event));   // every op is on this stream
            ReportErrors(memorypool_);
            env->IPCPost(local_dev_id_, current_pipe_);
            if (batch_id % 1000 == 0 && local_dev_id_ == 0) std::cout << "batch id: " << batch_id << "\n";
            current_pipe_ = (current_pipe_ + 1) % interbatch_concurrency_;
            memorypool_->SetCurrentPipe(current_pipe_);
            return;
        }
        if (pipe_ == nullptr) PrepareServing(params);
        const int32_t k = batch_id;
        const int p = current_pipe_;
        const auto t_a = std::chrono::steady_clock::now();
        {   // the trainer has released this slot?  Poll for a short while (a futex wake costs more than a small batch's
            // GPU time) before blocking: ~20000 polls with the gather hand-over, where the per-batch round trip IS the rate;
            // LegionTuning.runner_spin_us (20 us by default) once batches go out as views -- the trainer then sets the pace and a
            // group completes every few ms, so the thread sleeps instead of holding a core
            bool got = env->IPCTryWait(local_dev_id_, p);
            if (!got && kind_ == KIND_VIEWS && spin_us_ >= 0) {
                const auto t0 = std::chrono::steady_clock::now();
                while (!got && std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() < spin_us_)
                    got = env->IPCTryWait(local_dev_id_, p);
            } else {
                for (int spin = 0; spin < 20000 && !got; spin++) got = env->IPCTryWait(local_dev_id_, p);
            }
            if (!got) env->IPCWait(local_dev_id_, p);
        }
        const auto t_b = std::chrono::steady_clock::now();
        if (kind_ == KIND_UNDECIDED) DecideHandover(env);       // the trainer end said what it is before its first sem_post
        sched_.retire_before(k);
        SubmitWhatFits(env, k);                                  // token k consumed => batches <= k-2 are released (runner_schedule.h)
        if (sched_.groups.empty() || k < sched_.groups.front().first) { printf("legion_hip: runner lost track of its groups\n"); exit(EXIT_FAILURE); }
        RunnerSchedule::Group& g = sched_.groups.front();
        if (!g.complete) {
            if (kind_ == KIND_GATHER) {
                // the hand-over streams may touch this group's lanes once its sampler phase has finished
                for (int i = 0; i < 2; i++)
                    if (ho_streams_[i] != nullptr)
                        HIP_CALL(hipStreamWaitEvent(ho_streams_[i], (hipEvent_t)legion_pipeline_slot_done_event(pipe_, g.slot), 0));
            } else if (kind_ == KIND_VIEWS && spin_us_ >= 0) {
                legion_pipeline_wait_sleeping(pipe_, g.slot, spin_us_);    // the group has completed on the GPU (the thread slept meanwhile)
            } else {
                legion_pipeline_wait(pipe_, g.slot);             // the group has completed on the GPU
            }
            g.complete = true;
        }
        const int32_t lane = k - g.first;
        MemoryPool* lp = reinterpret_cast<MemoryPool*>(legion_pipeline_pool(pipe_, g.slot, lane));
        if (kind_ == KIND_VIEWS) {
            const LanePtrs h = lp->HostLane(0);
            int64_t off[5] = {(char*)h.sampled_ids - arena_.base, (char*)h.float_features - arena_.base, (char*)h.labels - arena_.base,
                              (char*)h.agg_src_off - arena_.base, (char*)h.agg_dst_off - arena_.base};
            // A batch with more rows than its lane's feature buffer (sized 1.2 x the largest TRAINING batch PreSC saw, server.cu:277:
            // validation / test seeds with heavier neighbourhoods exceed it) is not a reason to stop serving: its rows are
            // gathered once more, ALL of them, into the pipe slot's overflow buffer -- num_ids rows, inside the arena -- and the view of
            // its rows points there.  The slot's previous batch has been released (the token this call consumed), so the buffer is free.
            const int32_t batch_rows = lp->counter_mirror_host != nullptr ? lp->counter_mirror_host[INTRABATCH_CON * 3 + hop_num_] : 0;
            bool overflowed = false;
            if (float_feature_len_ > 0 && batch_rows > lane_feature_rows_ && overflow_[p] != nullptr) {
                UnifiedCache* cache = (UnifiedCache*)(params->cache);
                const LanePtrs* desc = d_overflow_desc_ + ((size_t)p * slots_ + g.slot) * lanes_ + lane;
                cache->FeatCacheLookup(desc, 1, INTRABATCH_CON * hop_num_ + 1, local_dev_id_, overflow_stream_, memorypool_->num_ids, true, 1);
                HIP_CALL(hipStreamSynchronize(overflow_stream_));
                off[1] = (char*)overflow_[p] - arena_.base;
                if (lp->err_host != nullptr) *(volatile int32_t*)lp->err_host &= ~LG_ERR_FEATURE_ROWS;      // this batch is whole again; the lane's next batch starts clean
                overflowed = true;
                overflow_batches_++;
                if (overflow_batches_ == 1)
                    std::cout << "WARNING (gpu " << local_dev_id_ << "): batch " << k << " has " << batch_rows << " rows, its lane's feature buffer "
                              << lane_feature_rows_ << " (1.2 x the PreSC maximum): its rows are handed over from the pipe slot's overflow buffer\n" << std::flush;
            }
            ReportErrors(lp, !overflowed);
            env->SetView(local_dev_id_, p, off, lp->counter_mirror_host);
            env->IPCPost(local_dev_id_, p);
        } else {
            // hand-overs run on their own (high-priority) streams, one per pipe slot: the two in flight overlap on the GPU
            hipStream_t s = ho_streams_[p % 2] != nullptr ? ho_streams_[p % 2] : static_cast<hipStream_t>(legion_pipeline_stream(pipe_));
            const int64_t max_rows = std::min<int64_t>(memorypool_->feature_rows, memorypool_->num_ids);
            const LanePtrs* desc = d_desc_ + ((size_t)p * slots_ + g.slot) * lanes_ + lane;     // lane -> pipe slot p
            UnifiedCache* cache = (UnifiedCache*)(params->cache);
            if (float_feature_len_ > 0 && max_rows > 0)   // one launch: gather of every row of the batch + the hand-over copies
                cache->FeatCacheLookup(desc, 1, INTRABATCH_CON * hop_num_ + 1, local_dev_id_, s, (int32_t)max_rows, true, 1, true, false, lane_rule_rows_);
            else
                lg::launch_deliver(s, desc, deliver_[p]);
            HIP_CALL(hipEventRecord(batch_done_[p], s));
            // single producer (this thread), single consumer (the poster): at most INTERBATCH_CON jobs are outstanding
            // (a slot is only reused after the trainer released it, i.e. after its previous job was posted)
            const uint32_t t = q_tail_.load(std::memory_order_relaxed);
            ring_[t % kRing] = {batch_done_[p], p, lp, std::chrono::steady_clock::now()};
            q_tail_.store(t + 1, std::memory_order_release);
        }
        if (stats_) {
            const auto t_c = std::chrono::steady_clock::now();
            st_wait_ += std::chrono::duration<double>(t_b - t_a).count();
            st_launch_ += std::chrono::duration<double>(t_c - t_b).count();
            st_n_++;
        }
        if (k % 1000 == 0 && local_dev_id_ == 0) std::cout << "batch id: " << k << "\n";
        current_pipe_ = (current_pipe_ + 1) % interbatch_concurrency_;
    }

    // Everything serving needs that allocates, captures or instantiates -- lanes, descriptors, every group shape's graphs, the
    // first groups' launches -- done BEFORE the server announces itself: while trainers attach to the IPC buffers the server
    // process then only launches kernels (concurrent allocation in the exporting process made hipIpcOpenMemHandle fail
    // now and then with two trainers attaching at once, ROCm 7.2), and the first batches are ready when the trainer asks.
    void PrepareServing(RunnerParams* params) override
    {
        if (!use_groups_ || pipe_ != nullptr) return;
        SetGPUDevice(local_dev_id_);
        IPCEnv* env = (IPCEnv*)(params->env);
        CreateGroups(params);
        // every group shape of the whole schedule gets its graphs now, with and without the gathers (which of the two runs is
        // known only when the trainer end attaches): no stream capture while the poster thread polls events
        for (int with_gathers = 0; with_gathers < 2; with_gathers++) {
            if ((handover_ == 1 || !lane_features_) && with_gathers) continue;     // forced `gather` (or no features at all): the groups never gather
            legion_pipeline_set_gathers(pipe_, with_gathers);
            for (int32_t first = 0; first < max_step_;) {
                int32_t mode = 0, local0 = 0;
                const int32_t n = PlanGroup(env, first, mode, local0);
                legion_pipeline_prepare(pipe_, mode, n, env->GetCurrentBatchsize(local_dev_id_, mode));
                first += n;
            }
        }
        // the first group runs now (with its gathers unless they are ruled out: a trainer end that takes views finds its first
        // batches ready, any other finds them sampled -- its rows are gathered at hand-over either way)
        legion_pipeline_set_gathers(pipe_, handover_ != 1 && lane_features_);
        if (max_step_ > 0) {
            SubmitNext(env);
            HIP_CALL(hipStreamSynchronize(static_cast<hipStream_t>(legion_pipeline_stream(pipe_))));
        }
    }

    void Finalize(RunnerParams* params) override
    {
        IPCEnv* env = (IPCEnv*)(params->env);
        SetGPUDevice(local_dev_id_);
        if (poster_.joinable()) {
            stop_.store(true, std::memory_order_release);
            poster_.join();                                                // every queued batch has been posted
        }
        env->IPCWait(local_dev_id_, (current_pipe_ + 1) % interbatch_concurrency_);
        SetGPUDevice(local_dev_id_);
        if (stats_ && st_n_ > 0)   // LEGION_RUNNER_STATS=1: where a hand-over's time goes, averages per batch
            std::cout << "runner " << local_dev_id_ << ": " << st_n_ << " hand-overs; waiting for a free slot " << st_wait_ / st_n_ * 1e6
                      << " us, hand-over calls " << st_launch_ / st_n_ * 1e6 << " us, enqueue -> completion seen "
                      << st_gpu_ / st_n_ * 1e6 << " us\n";
        for (int i = 0; i < 2; i++) {
            if (ho_streams_[i] == nullptr) continue;
            HIP_CALL(hipStreamSynchronize(ho_streams_[i]));
            if (i == 0 || ho_streams_[1] != ho_streams_[0]) HIP_CALL(hipStreamDestroy(ho_streams_[i]));
        }
        ho_streams_[0] = ho_streams_[1] = nullptr;
        if (pipe_) {
            legion_pipeline_destroy(pipe_);
            pipe_ = nullptr;
            if (arena_.base) d_free_space(arena_.base);
            if (arena_.mirror_host) HIP_CALL(hipHostFree(arena_.mirror_host));
            arena_ = PoolArena();
            d_free_space(d_desc_);
            d_desc_ = nullptr;
            d_free_space(d_overflow_desc_);
            d_overflow_desc_ = nullptr;
            if (overflow_stream_ != nullptr) { HIP_CALL(hipStreamDestroy(overflow_stream_)); overflow_stream_ = nullptr; }
            if (overflow_batches_ > 0) std::cout << "runner " << local_dev_id_ << ": " << overflow_batches_ << " batches handed over from an overflow buffer\n";
            d_free_space(d_deliver_);
            d_deliver_ = nullptr;
        }
        memorypool_->Finalize();
    }

private:
    struct Pending { hipEvent_t ev; int pipe; MemoryPool* lane; std::chrono::steady_clock::time_point enqueued; };
    enum { KIND_UNDECIDED = 0, KIND_VIEWS, KIND_GATHER };

    // as_view: the batch is about to be handed over as views of its lane
    void ReportErrors(MemoryPool* mp, bool as_view = false)
    {
        int32_t bits = mp->ErrorBits() & ~reported_;
        // a lane's own gather may have stopped at the end of the lane's (1.2 x rule) buffer although the batch then went out whole
        // through a pipe slot that holds the worst case: not an event for that hand-over
        if (!as_view && mp != memorypool_ && memorypool_->feature_rows >= memorypool_->num_ids) bits &= ~LG_ERR_FEATURE_ROWS;
        if (bits == 0) return;
        reported_ |= bits;
        if (bits & LG_ERR_FEATURE_ROWS)
            std::cout << (as_view ? "ERROR" : "WARNING") << " (gpu " << local_dev_id_ << "): a batch has more rows than the feature buffer ("
                      << (as_view ? (int64_t)lane_feature_rows_ : memorypool_->feature_rows) << " rows"
                      << (as_view || memorypool_->feature_rows < memorypool_->num_ids ? " = 1.2 x the PreSC maximum" : "") << "); its tail rows were not gathered\n";
        if (bits & LG_ERR_TABLE_FULL) std::cout << "ERROR (gpu " << local_dev_id_ << "): a de-duplication bucket fits no LDS table\n";
        if (bits & LG_ERR_CHAIN) std::cout << "ERROR (gpu " << local_dev_id_ << "): unresolved first-touch chain\n";
        std::cout << std::flush;
        // An unresolved bucket or chain means positions in this batch are garbage: it must not reach a trainer.
        // Same convention as every other device-side failure at this boundary (include/legion_hip.h, the reference's
        // cudaCheckError): say what happened and end the process -- the caller is about to IPCPost.  A truncated feature
        // buffer stays a warning where the rows were gathered into the pipe slot's own buffer (ids, edges and the rows that fit
        // are correct; the reference overruns here, SS/engine/server.cu:277) -- but a VIEW of a lane's feature buffer with more
        // rows than the buffer has would show the trainer the neighbouring lane's arrays as rows: never posted.
        if ((bits & (LG_ERR_TABLE_FULL | LG_ERR_CHAIN)) || (as_view && (bits & LG_ERR_FEATURE_ROWS))) {
            std::cout << "legion_hip: corrupt batch on gpu " << local_dev_id_ << ", not posted; stopping the server\n" << std::flush;
            // trainers blocked in sem_wait must not wait for ever: mark the mirror object (this build's trainer end checks it
            // after every wake-up and fails), wake every waiter, unlink the names -- then go (no destructors: other runner
            // threads are still in launch paths)
            if (env_ != nullptr) env_->AbortServing();
            _exit(EXIT_FAILURE);
        }
    }

    // what the trainer end of this GPU is, known when the first semaphore token arrives
    void DecideHandover(IPCEnv* env)
    {
        if (handover_ == 0 && env->TrainerTakesViews(local_dev_id_)) kind_ = KIND_VIEWS;
        else kind_ = KIND_GATHER;
        // groups submitted from now on gather their rows only when somebody reads them from the lanes
        legion_pipeline_set_gathers(pipe_, kind_ != KIND_GATHER && lane_features_);
        std::cout << "runner " << local_dev_id_ << ": hand-over by "
                  << (kind_ == KIND_VIEWS ? "views of the lane arena" : "one gather launch per batch into the pipe slots")
                  << ", " << lanes_ << " lanes per group, " << slots_ << " groups in flight\n" << std::flush;
    }

    // the largest group the schedule ever forms (consecutive batches of one mode with consecutive local ids)
    int32_t LargestGroup(IPCEnv* env, int32_t cap)
    {
        int32_t best = 1, keep = lanes_;
        lanes_ = cap;
        for (int32_t first = 0; first < max_step_;) {
            int32_t mode = 0, local0 = 0;
            const int32_t n = PlanGroup(env, first, mode, local0);
            best = std::max(best, n);
            first += n;
        }
        lanes_ = keep;
        return best;
    }

    // lanes x groups in flight, their arena, the hand-over descriptors of every (pipe slot, group slot, lane), the poster
    void CreateGroups(RunnerParams* params)
    {
        IPCEnv* env = (IPCEnv*)(params->env);
        hop_num_ = (int32_t)params->fanout.size();
        max_step_ = env->GetMaxStep();
        std::vector<int32_t> fanout(params->fanout.begin(), params->fanout.end());
        const LegionTuning tune = lg::tuning();
        const int64_t feature_rows = std::max<int64_t>(1, std::min<int64_t>(lane_rule_rows_, memorypool_->num_ids));     // the lanes: 1.2 x the PreSC maximum
        // groups as large as bench.py's (524288 / B rounded down to a power of two, at most 512: the launch tails and the
        // per-kernel floors are paid once per group), never larger than the schedule can fill, halved while the lanes of
        // the groups in flight would take more than 0.6 of the HBM that is free now (tables and caches are in place).
        // Three groups in flight: one being handed over, one running, one queued behind it -- with two, the GPU could only
        // start group g+2 after the trainer had taken the first batch of g+1, and the next group's head never ran under the
        // current group's heavy kernels (measured at RMAT-26, views: 132 k batches/s with two)
        slots_ = tune.runner_slots >= 2 ? std::min(tune.runner_slots, 4) : 3;
        lanes_ = 1;
        while (lanes_ * 2 <= 512 && (int64_t)lanes_ * 2 * memorypool_->batch_size <= 524288) lanes_ *= 2;
        if (tune.runner_lanes > 0) lanes_ = tune.runner_lanes;
        lanes_ = std::min(lanes_, LargestGroup(env, lanes_));
        const bool lane_features = handover_ != 1 && float_feature_len_ > 0;        // forced `gather`: rows never land in a lane
        lane_features_ = lane_features;
        const int64_t arena_lane = lg_pool_arena_bytes(memorypool_->batch_size, memorypool_->num_ids, lane_features ? feature_rows : 0, float_feature_len_);
        const int64_t lane_bytes = arena_lane + (int64_t)memorypool_->num_ids * 40 + (int64_t)memorypool_->max_slots * 28;
        size_t free_b = 0, total_b = 0;
        HIP_CALL(hipMemGetInfo(&free_b, &total_b));
        while (lanes_ > 1 && (int64_t)lanes_ * slots_ * lane_bytes > (int64_t)(free_b / 10 * 6)) lanes_ /= 2;
        arena_.bytes = arena_lane * lanes_ * slots_;
        lane_feature_rows_ = lane_features ? (int32_t)feature_rows : 0;
        // one overflow buffer per pipe slot behind the lanes, inside the arena (a trainer end that takes views maps the whole arena): the
        // worst case of a batch, num_ids rows -- unless two of them would not leave the lanes their memory (then an oversized batch
        // stops the server as before)
        int64_t overflow_bytes = 0;
        if (lane_features && handover_ == 0 && feature_rows < memorypool_->num_ids && tune.runner_overflow != 0) {
            overflow_bytes = (((int64_t)memorypool_->num_ids * float_feature_len_ * (int64_t)sizeof(float)) + 4095) & ~(int64_t)4095;
            if (interbatch_concurrency_ * overflow_bytes > (int64_t)(free_b / 10) || (int64_t)lanes_ * slots_ * lane_bytes + interbatch_concurrency_ * overflow_bytes > (int64_t)(free_b / 10 * 7))
                overflow_bytes = 0;
        }
        arena_.bytes += interbatch_concurrency_ * overflow_bytes;
        // The arena is built from physical chunks mapped in shuffled order unless another GPU must reach it (peer_gather = bulk):
        // 128 MB chunks -- a trainer end receives them as file descriptors (IPCEnv::PublishArena) -- against the 2 MB of an arena
        // nobody else maps; LegionTuning.arena_scatter_mb = 0: one plain allocation, handed over as a hipIpcMemHandle.
        {
            const int32_t mb = tune.arena_scatter_mb;
            const bool plain = mb <= 0;      // (peer_gather = bulk: the other GPUs of the clique are granted access below, bulk_enable_shared)
            arena_.base = (char*)(plain ? d_alloc_space(arena_.bytes) : d_alloc_scattered_exportable(arena_.bytes, std::max(mb, 128)));
        }
        arena_.used = 0;
        arena_.mirror_lanes = lanes_ * slots_;
        arena_.mirror_used = 0;
        HIP_CALL(hipHostMalloc((void**)&arena_.mirror_host, (size_t)arena_.mirror_lanes * 32 * sizeof(int32_t), hipHostMallocMapped));
        memset(arena_.mirror_host, 0, (size_t)arena_.mirror_lanes * 32 * sizeof(int32_t));
        HIP_CALL(hipHostGetDevicePointer((void**)&arena_.mirror_dev, arena_.mirror_host, 0));
        lg_set_pool_arena(&arena_);
        // use_graph bits: 1 graph replay, 16 weave (the next group's head on a second stream under this group's heavy kernels)
        pipe_ = legion_pipeline_create((LegionGraphStorage*)params->graph, (LegionFeatureStorage*)params->feature,
                                       (LegionUnifiedCache*)params->cache, local_dev_id_, memorypool_->batch_size,
                                       fanout.data(), hop_num_, lanes_, slots_, lane_features ? feature_rows : 0, 1 | 16);
        lg_set_pool_arena(nullptr);
        for (int pp = 0; pp < interbatch_concurrency_ && pp < INTERBATCH_CON; pp++)
            overflow_[pp] = overflow_bytes > 0 ? (float*)(arena_.base + arena_.bytes - (int64_t)(interbatch_concurrency_ - pp) * overflow_bytes) : nullptr;
        if (overflow_bytes > 0) {
            if (arena_.used > arena_.bytes - interbatch_concurrency_ * overflow_bytes) { printf("legion_hip: the lanes overran their share of the arena\n"); exit(EXIT_FAILURE); }
            HIP_CALL(hipStreamCreateWithFlags(&overflow_stream_, hipStreamNonBlocking));
        }
        {   // LegionTuning.peer_gather = bulk: only where it means something (a clique of several GPUs, rows landing in lanes)
            UnifiedCache* uc = (UnifiedCache*)(params->cache);
            bulk_ = tune.peer_gather == 1 && uc->Kg_ > 1 && lane_features && legion_pipeline_bulk_enable_shared(pipe_, &arena_) != 0;
            if (bulk_) std::cout << "runner " << local_dev_id_ << ": rows of other members' stripes are pushed by their owners (peer_gather = bulk)\n";
        }
        if (handover_ == 0 && lane_features && env->PublishArena(local_dev_id_, arena_.base, arena_.bytes))
            std::cout << "runner " << local_dev_id_ << ": lane arena of " << (arena_.bytes >> 20) << " MiB published ("
                      << lanes_ << " lanes x " << slots_ << " groups)\n" << std::flush;
        sched_.reset(slots_);
        const int32_t ho_mode = tune.runner_ho_stream;    // 0 the pipeline's stream, 1 one shared, 2 one per pipe slot
        if (ho_mode != 0) {
            int lo = 0, hi = 0;
            HIP_CALL(hipDeviceGetStreamPriorityRange(&lo, &hi));
            HIP_CALL(hipStreamCreateWithPriority(&ho_streams_[0], hipStreamNonBlocking, hi));
            // one stream per pipe slot: the two hand-overs in flight then overlap on the GPU (the tail of one small launch
            // under the head of the next) instead of queueing -- 57.9 k against 47.9 k batches/s at B = 1024, 15.5 k against
            // 13.2 k at B = 8000 (runner_ho_stream = 1: one stream for both)
            if (ho_mode == 1)
                ho_streams_[1] = ho_streams_[0];
            else
                HIP_CALL(hipStreamCreateWithPriority(&ho_streams_[1], hipStreamNonBlocking, hi));
        }
        for (int p = 0; p < interbatch_concurrency_; p++) {
            lg::DeliverParams& d = deliver_[p];
            d.sampled_ids = env->GetIds(local_dev_id_, p);
            d.labels = env->GetLabels(local_dev_id_, p);
            d.agg_src_off = env->GetAggSrc(local_dev_id_, p);
            d.agg_dst_off = env->GetAggDst(local_dev_id_, p);
            d.node_counter = env->GetNodeCounter(local_dev_id_, p);
            d.edge_counter = env->GetEdgeCounter(local_dev_id_, p);
            d.mirror = env->GetCounterMirror(local_dev_id_, p);
            d.num_ids = memorypool_->num_ids;
            d.batch_cap = memorypool_->batch_size;
            HIP_CALL(hipEventCreateWithFlags(&batch_done_[p], hipEventDisableTiming));
        }
        d_deliver_ = (lg::DeliverParams*)d_alloc_space(sizeof(deliver_));
        HIP_CALL(hipMemcpy(d_deliver_, deliver_, sizeof(deliver_), hipMemcpyHostToDevice));
        // `gather` hand-over descriptors [pipe slot][group slot][lane]: the lane's own buffers as the source, the pipe slot's
        // feature buffer as the gather's destination, its other arrays as deliver_slice's
        std::vector<LanePtrs> h((size_t)interbatch_concurrency_ * slots_ * lanes_);
        for (int p = 0; p < interbatch_concurrency_; p++)
            for (int sl = 0; sl < slots_; sl++)
                for (int g = 0; g < lanes_; g++) {
                    MemoryPool* lp = reinterpret_cast<MemoryPool*>(legion_pipeline_pool(pipe_, sl, g));
                    LanePtrs d = lp->HostLane(0);
                    d.float_features = env->GetFloatFeatures(local_dev_id_, p);
                    d.feature_rows = (int32_t)std::min<int64_t>(memorypool_->feature_rows, memorypool_->num_ids);
                    d.deliver = d_deliver_ + p;
                    // counters stay the lane's: deliver_slice reads them there and writes the slot's (incl. [2..3])
                    h[((size_t)p * slots_ + sl) * lanes_ + g] = d;
                }
        d_desc_ = (LanePtrs*)d_alloc_space((int64_t)h.size() * sizeof(LanePtrs));
        HIP_CALL(hipMemcpy(d_desc_, h.data(), h.size() * sizeof(LanePtrs), hipMemcpyHostToDevice));
        if (overflow_[0] != nullptr) {      // the same lanes with the pipe slot's overflow buffer as the gather's destination, nothing delivered
            for (int p = 0; p < interbatch_concurrency_; p++)
                for (size_t i = 0; i < (size_t)slots_ * lanes_; i++) {
                    LanePtrs& d = h[(size_t)p * slots_ * lanes_ + i];
                    d.float_features = overflow_[p];
                    d.feature_rows = memorypool_->num_ids;
                    d.deliver = nullptr;
                }
            d_overflow_desc_ = (LanePtrs*)d_alloc_space((int64_t)h.size() * sizeof(LanePtrs));
            HIP_CALL(hipMemcpy(d_overflow_desc_, h.data(), h.size() * sizeof(LanePtrs), hipMemcpyHostToDevice));
        }
        poster_ = std::thread([this, env] {
            // The poster polls (the reference's runner thread polls cudaEventQuery the same way, server.cu:319-324): one core
            // per GPU buys a hand-over latency of about a microsecond instead of a condition-variable wake-up.
            // (Measured and rejected, tools/micro/stream_wait_probe.hip + LEGION_RUNNER_STATS: queueing the hand-over ahead
            // behind hipStreamWaitValue32 and signalling completion with hipStreamWriteValue32 -- 36 k batches/s against
            // 44 k with events at B = 1024.)  With the `views` hand-over it has nothing to do and sleeps.
            SetGPUDevice(local_dev_id_);
            {
                char name[16];
                snprintf(name, sizeof(name), "lg-poster%d", local_dev_id_);
                pthread_setname_np(pthread_self(), name);
            }
            uint32_t head = 0;
            for (uint32_t idle = 0;;) {
                if (q_tail_.load(std::memory_order_acquire) == head) {
                    if (stop_.load(std::memory_order_acquire) && q_tail_.load(std::memory_order_acquire) == head) return;
                    ++idle;
                    if (idle > (1u << 18)) std::this_thread::sleep_for(std::chrono::microseconds(50));   // trainer is slow: stop burning the core
                    else if ((idle & 4095) == 0) std::this_thread::yield();
                    continue;
                }
                const Pending job = ring_[head % kRing];
                const hipError_t q = hipEventQuery(job.ev);
                if (q == hipErrorNotReady) continue;
                if (q != hipSuccess) {
                    printf("HIP failure %s:%d: '%s'\n", __FILE__, __LINE__, hipGetErrorString(q));
                    exit(EXIT_FAILURE);
                }
                if (stats_) st_gpu_ += std::chrono::duration<double>(std::chrono::steady_clock::now() - job.enqueued).count();
                ReportErrors(job.lane);
                env->IPCPost(local_dev_id_, job.pipe);
                head++;
                idle = 0;
            }
        });
    }

    // the group that starts at global batch `first`: consecutive batches of one mode with consecutive local ids
    // (ipc_service.cu:213-253), at most `lanes_`
    int32_t PlanGroup(IPCEnv* env, int32_t first, int32_t& mode, int32_t& local0)
    {
        mode = env->GetCurrentMode(first);
        local0 = env->GetLocalBatchId(first);
        int32_t n = 1;
        while (n < lanes_ && first + n < max_step_ && env->GetCurrentMode(first + n) == mode &&
               env->GetLocalBatchId(first + n) == local0 + n)
            n++;
        return n;
    }

    // enqueues the next group of the schedule on the pipeline's next slot
    void SubmitNext(IPCEnv* env)
    {
        int32_t mode = 0, local0 = 0;
        const int32_t n = PlanGroup(env, sched_.next_first, mode, local0);
        const int target = sched_.next_slot();
        // peer_gather = bulk (striped caches, rows landing in the lanes): the rows of other members' stripes are listed per owner
        // and pushed by kernels on the owners' devices instead of being loaded one by one over xGMI (pipeline.hip)
        const bool bulk = bulk_ && kind_ != KIND_GATHER && kind_ != KIND_UNDECIDED;
        const int slot = bulk ? legion_pipeline_submit_bulk_inproc(pipe_, local0, mode, n, env->GetCurrentBatchsize(local_dev_id_, mode))
                              : legion_pipeline_submit_ex(pipe_, local0, mode, n, env->GetCurrentBatchsize(local_dev_id_, mode));
        if (slot != target) { printf("legion_hip: runner lost track of the pipeline's slots\n"); exit(EXIT_FAILURE); }
        sched_.submitted(slot, n);
    }

    // Keeps up to slots_ groups submitted; a slot is overwritten only when the trainer has released every batch of the group
    // that used it last (runner_schedule.h: the rule and why the semaphore tokens are enough to know)
    void SubmitWhatFits(IPCEnv* env, int32_t k)
    {
        while (sched_.next_first < max_step_ && sched_.may_submit(k)) SubmitNext(env);
    }

    int32_t num_ids_ = 0;
    int32_t float_feature_len_ = 0;
    MemoryPool* memorypool_ = nullptr;
    IPCEnv* env_ = nullptr;
    int current_pipe_ = 0;
    int interbatch_concurrency_ = INTERBATCH_CON;
    int local_dev_id_ = 0;
    int mode_ = 0;
    int op_num_ = 0;
    std::vector<hipStream_t> streams_;
    std::vector<hipEvent_t> events_;
    std::vector<Operator*> op_factory_;
    std::vector<OpParams*> op_params_;
    bool use_groups_ = lg::tuning().runner_graph != 0;
    // launch groups
    int32_t spin_us_ = lg::tuning().runner_spin_us < 0 ? 20 : lg::tuning().runner_spin_us;      // views hand-over: poll this long, then sleep (< 0 in the struct: the default)
    int32_t handover_ = lg::tuning().runner_handover;     // 0 auto (views, else gather), 1 gather
    int kind_ = KIND_UNDECIDED;
    bool lane_features_ = false;                           // the lanes have feature buffers (and the groups may gather into them)
    bool bulk_ = false;                                    // peer_gather = bulk is in effect for this runner's groups
    LegionPipeline* pipe_ = nullptr;
    PoolArena arena_;
    int32_t lanes_ = 1, slots_ = 3, hop_num_ = 0, max_step_ = 0;
    RunnerSchedule sched_;                                 // which group sits in which slot, and when a slot may be overwritten
    LanePtrs* d_desc_ = nullptr;
    LanePtrs* d_overflow_desc_ = nullptr;                  // views: [pipe slot][group slot][lane] with the slot's overflow buffer as the rows' destination
    float* overflow_[INTERBATCH_CON] = {};                 // per pipe slot: num_ids rows inside the arena, for a batch that outgrew its lane's buffer
    hipStream_t overflow_stream_ = nullptr;
    int32_t lane_feature_rows_ = 0;
    int32_t lane_rule_rows_ = 1;                           // 1.2 x the PreSC maximum (scaled by the largest batch of any mode): what a lane's feature buffer holds
    int64_t overflow_batches_ = 0;
    lg::DeliverParams deliver_[INTERBATCH_CON] = {};
    lg::DeliverParams* d_deliver_ = nullptr;
    hipEvent_t batch_done_[INTERBATCH_CON] = {};
    int32_t reported_ = 0;
    std::thread poster_;
    static constexpr uint32_t kRing = 8;
    Pending ring_[kRing] = {};
    std::atomic<uint32_t> q_tail_{0};
    std::atomic<bool> stop_{false};
    hipStream_t ho_streams_[2] = {nullptr, nullptr};    // hand-over launches of pipe slot 0 / 1 (LEGION_RUNNER_HO_STREAM=0: the pipeline's stream; 1: one for both)
    bool stats_ = lg::tuning().runner_stats != 0;
    double st_wait_ = 0, st_launch_ = 0, st_gpu_ = 0;
    int64_t st_n_ = 0;
};

Runner* NewGPURunner() { return new GPURunner(); }

// =============================================================================================
static void PreSCLoop(int train_step, Runner* runner, RunnerParams* params)
{
    for (int i = 0; i < train_step; i++) {
        params->global_batch_id = i;
        runner->RunPreSc(params);
    }
    runner->InitializeFeaturesBuffer(params);
}

static void RunnerLoop(int max_step, Runner* runner, RunnerParams* params)
{
    {   // (thread names show in /proc/<pid>/task/*/comm: tools/server_throughput.py reports the server's CPU time by thread)
        char name[16];
        snprintf(name, sizeof(name), "lg-runner%d", params->device_id);
        pthread_setname_np(pthread_self(), name);
    }
    for (int i = 0; i < max_step; i++) {
        params->global_batch_id = i;
        runner->RunOnce(params);
    }
}

class GPUServer : public Server {
public:
    void Initialize(int global_shard_count, std::vector<int> fanout, int in_memory_mode) override
    {
        shard_count_ = global_shard_count;
        in_memory_mode_ = in_memory_mode;
        lg::tuning_refresh();           // the environment as it is when the server starts (or what the host program installed)
        if (in_memory_mode) std::cout << "In Memory Mode\n";
        else std::cout << "In Disk Mode\n";
        StorageManagement* storage_management = new StorageManagement();
        if (!storage_management->Initialze(shard_count_, in_memory_mode)) exit(EXIT_FAILURE);
        graph_ = storage_management->GetGraph();
        feature_ = storage_management->GetFeature();
        cache_ = storage_management->GetCache();
        ipc_env_ = storage_management->GetIPCEnv();
        train_step_ = ipc_env_->GetTrainStep();
        max_step_ = ipc_env_->GetMaxStep();
        runners_.resize(shard_count_);
        params_.resize(shard_count_);
        for (int i = 0; i < shard_count_; i++) {
            SetGPUDevice(i);
            RunnerParams* p = new RunnerParams();
            p->device_id = i;
            p->fanout = fanout;
            p->cache = (void*)cache_;
            p->graph = (void*)graph_;
            p->feature = (void*)feature_;
            p->env = (void*)ipc_env_;
            p->global_batch_id = 0;
            p->in_memory = 1;
            params_[i] = p;
            runners_[i] = NewGPURunner();
            runners_[i]->Initialize(params_[i]);
        }
    }

    // SS/engine/server.cu:90-117
    void PreSc(int cache_agg_mode) override
    {
        std::chrono::steady_clock::time_point t1 = std::chrono::steady_clock::now();
        const LegionTuning& tune = lg::tuning();
        const bool use_smi = tune.link_counters == 2;
        LinkTotals before, after;
        if (use_smi && !ReadLinkTotals(before, true)) {
            // the link's own counters were asked for: a table this build cannot decode is an error, not a reason to
            // quietly size the caches from something else
            std::cout << "legion_hip: LEGION_LINK_COUNTERS=smi but the driver's gpu_metrics table is missing or has an unknown "
                         "revision (known: 1.8); use LEGION_LINK_COUNTERS=measured for the counts the kernels compute\n" << std::flush;
            exit(EXIT_FAILURE);
        }
        std::vector<std::thread> pool;
        for (int i = 0; i < shard_count_; i++) pool.emplace_back(&PreSCLoop, train_step_, runners_[i], params_[i]);
        for (auto& th : pool) th.join();
        if (use_smi) {
            for (int i = 0; i < shard_count_; i++) { SetGPUDevice(i); HIP_CALL(hipDeviceSynchronize()); }
            std::this_thread::sleep_for(std::chrono::milliseconds(20));   // the table is refreshed every millisecond or so
            if (!ReadLinkTotals(after, false)) { std::cout << "legion_hip: gpu_metrics table vanished during PreSC\n" << std::flush; exit(EXIT_FAILURE); }
        }
        // PCIe/xGMI transaction counters of the PreSC epoch (Intel PCM in the paper, hard-wired to {0,0} in v2,
        // server.cu:105-106; CostModel sums the two into "transactions of topology", cache.cu:459).  LegionTuning.link_counters:
        //   v2       {0, 0}
        //   smi      [0] what the PCIe links of the server's GPUs carried during the epoch, [1] what those GPUs READ over
        //            their xGMI links, both in 64-byte transactions from the driver's cumulative counters (link_counters.hip)
        //   measured [0] the 64-byte topology transactions the sampler itself counted (legion_cache_topo_transactions),
        //            [1] the rows the gathers read from other members' stripes x row bytes / 64
        //            (legion_cache_peer_transactions; 0 in a fresh server -- PreSC runs no gathers -- non-zero when the
        //            cost model is re-run over a served epoch through the C API)
        //   "a,b"    injected values
        std::vector<uint64_t> counters(2, 0);
        if (use_smi) {
            counters[0] = (after.pcie - before.pcie) / 64;
            counters[1] = (after.xgmi_read - before.xgmi_read) / 64;
            std::cout << "PCIe transactions (gpu_metrics): " << counters[0] << "\n";
            std::cout << "xGMI read transactions (gpu_metrics): " << counters[1] << "\n";
        } else if (tune.link_counters == 1) {
            for (int i = 0; i < shard_count_; i++) {
                unsigned long long v = 0;
                SetGPUDevice(i);
                HIP_CALL(hipDeviceSynchronize());
                HIP_CALL(hipMemcpy(&v, cache_->Controller(i)->GetTopoTransactions(), sizeof(v), hipMemcpyDeviceToHost));
                counters[0] += v;
                counters[1] += legion_cache_peer_transactions((LegionUnifiedCache*)cache_, i);
            }
            std::cout << "Topology transactions: " << counters[0] << "\n";
            std::cout << "Peer-stripe transactions: " << counters[1] << "\n";
        } else if (tune.link_counters == 3) {
            counters[0] = tune.link_counter_values[0];
            counters[1] = tune.link_counter_values[1];
        }
        double t = std::chrono::duration_cast<std::chrono::duration<double>>(std::chrono::steady_clock::now() - t1).count();
        if (!in_memory_mode_ && cache_->CPUCapacity() + cache_->GPUCapacity() > 0) {
            // disk mode with the two capacities given: the hybrid CPU-cache / GPU-cache tier -- the call the reference keeps
            // commented out (server.cu:112) beside the three above it
            cache_->HybridInit(feature_, graph_);
        } else {
            cache_->CandidateSelection(cache_agg_mode, feature_, graph_);
            cache_->CostModel(cache_agg_mode, feature_, graph_, counters, train_step_);
            cache_->FillUp(cache_agg_mode, feature_, graph_);
        }
        for (int i = 0; i < shard_count_; i++) {
            params_[i]->global_batch_id = 0;
            runners_[i]->PrepareServing(params_[i]);
        }
        std::cout << "Preprocessing cost: " << t << " s\n";
        std::cout << "System is ready for serving\n" << std::flush;
    }

    // sums of the cumulative link counters of the (distinct physical) GPUs this server drives
    struct LinkTotals { uint64_t pcie = 0, xgmi_read = 0; };
    bool ReadLinkTotals(LinkTotals& t, bool log)
    {
        t = LinkTotals();
        const int physical = std::max(1, legion_device_count());
        for (int i = 0; i < std::min(shard_count_, physical); i++) {
            LegionLinkCounters c;
            const int32_t ok = legion_link_counters_ex(i, &c);
            if (log)
                std::cout << "gpu_metrics of gpu " << i << " (" << c.pci_bus_id << "): revision " << c.format_revision << "."
                          << c.content_revision << (ok ? "" : " -- not a layout this build knows") << "\n";
            if (!ok) return false;
            t.pcie += c.pcie_bytes;
            t.xgmi_read += c.xgmi_read_bytes;
        }
        return true;
    }

    void Run() override
    {
        std::vector<std::thread> pool;
        for (int i = 0; i < shard_count_; i++) pool.emplace_back(&RunnerLoop, max_step_, runners_[i], params_[i]);
        for (auto& th : pool) th.join();
    }

    void Finalize() override
    {
        for (int i = 0; i < shard_count_; i++) runners_[i]->Finalize(params_[i]);
        graph_->Finalize();
        feature_->Finalize();
        ipc_env_->Finalize();
        std::cout << "Server Stopped\n";
    }

private:
    GraphStorage* graph_ = nullptr;
    FeatureStorage* feature_ = nullptr;
    UnifiedCache* cache_ = nullptr;
    IPCEnv* ipc_env_ = nullptr;
    int shard_count_ = 0, train_step_ = 0, max_step_ = 0, in_memory_mode_ = 1;
    std::vector<Runner*> runners_;
    std::vector<RunnerParams*> params_;
};

// ---- C API ----------------------------------------------------------------------------------
extern "C" LegionServer* NewGPUServer(void) { return reinterpret_cast<LegionServer*>(static_cast<Server*>(new GPUServer())); }

extern "C" void legion_server_initialize(LegionServer* s, int32_t global_shard_count, const int32_t* fanout,
                                         int32_t hop_num, int32_t in_memory_mode)
{
    std::vector<int> f(fanout, fanout + hop_num);
    reinterpret_cast<Server*>(s)->Initialize(global_shard_count, f, in_memory_mode);
}
extern "C" void legion_server_presc(LegionServer* s, int32_t cache_agg_mode) { reinterpret_cast<Server*>(s)->PreSc(cache_agg_mode); }
extern "C" void legion_server_run(LegionServer* s) { reinterpret_cast<Server*>(s)->Run(); }
extern "C" void legion_server_finalize(LegionServer* s) { reinterpret_cast<Server*>(s)->Finalize(); }

// sampling_server/sampling_server.cpp:7-15
extern "C" int32_t legion_run(const int32_t* fanout, int32_t hop_num, int32_t gpu_number, int32_t in_memory_mode,
                              int32_t cache_mode)
{
    std::cout << "Start Sampling Server\n";
    LegionServer* server = NewGPUServer();
    legion_server_initialize(server, gpu_number, fanout, hop_num, in_memory_mode);
    legion_server_presc(server, cache_mode);
    legion_server_run(server);
    legion_server_finalize(server);
    return 0;
}
