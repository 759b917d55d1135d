// storage.hip -- GraphStorage / FeatureStorage / MemoryPool for the MI355X build.
//
// Reference: SS/storage/graph_storage.cu:12-111 (pointer tables with P+1 slots, slot P = full
// CSR, slot d = GPU d's cached CSR; GraphCache), SS/storage/feature_storage.cu:18-90 (per-GPU
// seed/label arrays), SS/engine/server.cu:216-234 + SS/engine/ipc_service.cu:134-211 (buffers),
// SS/engine/server_imp.cuh:2-51 (alloc helpers).
//
// Layout decision for 288 GB HBM: the "host" tier pointers (full CSR, full feature table) are just
// device-dereferenceable pointers; the caller places them in HBM when they fit and in mapped
// pinned memory otherwise.  Nothing here assumes which.
#include "legion_core.h"
#include <unistd.h>
#include <algorithm>
#include <map>
#include <mutex>
#include <vector>

#include <cstring>
#include <cerrno>
#include <cstddef>
#include <string>
#include <thread>
#include <sys/socket.h>
#include <sys/time.h>
#include <sys/un.h>

#include "../trainer/vmm_probe.h"      // vmm_import_fd(): imports a received descriptor whichever way THIS process's HIP runtime takes one

// ---- device helpers: logical device ids beyond the physical count map round-robin onto the
//      physical GPUs, so that clique striping (Kg > 1) can be exercised on a 1-GPU box ----------
static int lg_physical_count()
{
    static int n = -1;
    if (n < 0) {
        if (hipGetDeviceCount(&n) != hipSuccess || n < 1) {
            printf("legion_hip: no HIP device available -- this library has no CPU fallback\n");
            exit(EXIT_FAILURE);
        }
    }
    return n;
}

// One process per GPU (torchrun): the process's logical GPU 0 is physical GPU `base` (= LOCAL_RANK)
static int g_device_base = 0;
extern "C" void legion_set_device_base(int32_t base) { g_device_base = base < 0 ? 0 : base; }
extern "C" int32_t legion_get_device_base(void) { return g_device_base; }

static int g_local_only = -1;
extern "C" void legion_set_local_device(int32_t dev) { g_local_only = dev; }
bool lg_is_local(int32_t dev) { return g_local_only < 0 || dev == g_local_only; }

// physical device a logical GPU of this process runs on
int lg_physical_device(int32_t dev) { return (g_device_base + dev) % lg_physical_count(); }

extern "C" void SetGPUDevice(int32_t shard_id)
{
    HIP_CALL(hipSetDevice((g_device_base + shard_id) % lg_physical_count()));
}

extern "C" int32_t GetGPUDevice()
{
    int32_t dev_id = -1;
    HIP_CALL(hipGetDevice(&dev_id));
    return dev_id;
}

// ---- device memory whose pieces come from all over the HBM ----------------------------------------------------------------------------
// One contiguous VIRTUAL range backed by physical chunks that are created one by one and mapped in shuffled order (HIP virtual memory
// management).  Why ("where the lanes sit"): the last hop's gather writes a launch group's rows into the group's lanes; with
// the lanes in one hipMalloc'ed block -- one contiguous physical range -- it runs at 0.80 of the HBM peak, with the same block built from
// shuffled 2 MB ... 128 MB chunks at 0.87 (whole job 5.19 -> 5.57 G edges/s), better than what separate allocations get on a machine
// whose free memory happens to be fragmented (0.84-0.85) and independent of that luck.  Not exportable with hipIpcGetMemHandle and
// mapped for THIS device only; what another process or another GPU must reach is created exportable (below) and granted / served.
namespace {
struct ScatterLive {
    size_t bytes;
    std::vector<hipMemGenericAllocationHandle_t> chunks;
    std::vector<int> listeners;       // sockets of lg_scattered_serve threads (each thread takes its own out again when it ends)
    bool exportable = false;          // chunks created with a POSIX-file-descriptor handle type: only these can be exported / served
};
std::mutex g_scatter_mu;
std::map<void*, ScatterLive> g_scatter_live;
}
static void* alloc_scattered(int64_t num_bytes, int32_t chunk_mb, bool exportable);
extern "C" void* d_alloc_scattered(int64_t num_bytes, int32_t chunk_mb) { return alloc_scattered(num_bytes, chunk_mb, false); }
// ... whose chunks can be handed to another process as file descriptors (the server's lane arena: lg_scattered_export_fd)
extern "C" void* d_alloc_scattered_exportable(int64_t num_bytes, int32_t chunk_mb) { return alloc_scattered(num_bytes, chunk_mb, true); }
static void* alloc_scattered(int64_t num_bytes, int32_t chunk_mb, bool exportable)
{
    int dev = 0;
    HIP_CALL(hipGetDevice(&dev));
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    if (exportable) prop.requestedHandleType = hipMemHandleTypePosixFileDescriptor;
    size_t g0 = 0;
    HIP_CALL(hipMemGetAllocationGranularity(&g0, &prop, hipMemAllocationGranularityRecommended));          // (4 KB on this stack)
    const size_t want = (size_t)(chunk_mb > 0 ? chunk_mb : 2) << 20;
    const size_t g = (want + g0 - 1) / g0 * g0;
    const size_t n = ((size_t)(num_bytes > 0 ? num_bytes : 16) + g - 1) / g;
    ScatterLive live;
    live.bytes = n * g;
    live.exportable = exportable;
    live.chunks.resize(n);
    for (size_t i = 0; i < n; i++) HIP_CALL(hipMemCreate(&live.chunks[i], g, &prop, 0));
    uint64_t r = 0x9E3779B97F4A7C15ull;
    for (size_t i = n - 1; i > 0; i--) {                                                       // Fisher-Yates, fixed seed
        r ^= r << 13; r ^= r >> 7; r ^= r << 17;
        std::swap(live.chunks[i], live.chunks[(size_t)(r % (uint64_t)(i + 1))]);
    }
    void* ptr = nullptr;
    HIP_CALL(hipMemAddressReserve(&ptr, n * g, 0, nullptr, 0));
    for (size_t i = 0; i < n; i++) HIP_CALL(hipMemMap((char*)ptr + i * g, g, 0, live.chunks[i], 0));
    hipMemAccessDesc acc = {};
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = dev;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    HIP_CALL(hipMemSetAccess(ptr, n * g, &acc, 1));
    std::lock_guard<std::mutex> lk(g_scatter_mu);
    g_scatter_live[ptr] = std::move(live);
    return ptr;
}
// chunk size of a scattered allocation (0: `ptr` is not one) and its number of chunks, in mapping order
extern "C" int64_t lg_scattered_info(void* ptr, int32_t* n_chunks)
{
    std::lock_guard<std::mutex> lk(g_scatter_mu);
    auto it = g_scatter_live.find(ptr);
    if (it == g_scatter_live.end() || it->second.chunks.empty()) return 0;
    if (n_chunks) *n_chunks = (int32_t)it->second.chunks.size();
    return (int64_t)(it->second.bytes / it->second.chunks.size());
}
// a new file descriptor for chunk `index` of an exportable scattered allocation (the caller closes it), or -1.  The export runs
// under the registry's lock: d_free_scattered takes the allocation out of the registry under the same lock before it releases the
// chunks, so a handle is never exported while it is being released.
extern "C" int lg_scattered_export_fd(void* ptr, int32_t index)
{
    std::lock_guard<std::mutex> lk(g_scatter_mu);
    auto it = g_scatter_live.find(ptr);
    if (it == g_scatter_live.end() || !it->second.exportable || index < 0 || (size_t)index >= it->second.chunks.size()) return -1;
    int fd = -1;
    if (hipMemExportToShareableHandle(&fd, it->second.chunks[(size_t)index], hipMemHandleTypePosixFileDescriptor, 0) != hipSuccess) { (void)hipGetLastError(); return -1; }
    return fd;
}
// 1 when `ptr` is a scattered allocation whose chunks can be exported (d_alloc_scattered_exportable)
extern "C" int32_t lg_scattered_exportable(void* ptr)
{
    std::lock_guard<std::mutex> lk(g_scatter_mu);
    auto it = g_scatter_live.find(ptr);
    return (it != g_scatter_live.end() && it->second.exportable) ? 1 : 0;
}
// ---- a scattered allocation reached from OTHER GPUs / processes -------------------------------------------------------------------
// (the server's lane arena handed to a trainer; since round 5 also the lane arenas owners push rows into, peer_gather = bulk: round 4
// kept those plain because hipIpcGetMemHandle cannot export memory made with hipMemCreate, and the bulk leg started 7 % behind what
// the shuffled placement is worth to the gathers.)
// Same process, other GPUs: the range is made accessible to those devices as well.
extern "C" int32_t lg_scattered_grant(void* ptr, const int32_t* logical_devs, int32_t n)
{
    size_t bytes = 0;
    {
        std::lock_guard<std::mutex> lk(g_scatter_mu);
        auto it = g_scatter_live.find(ptr);
        if (it == g_scatter_live.end()) return 0;
        bytes = it->second.bytes;
    }
    std::vector<hipMemAccessDesc> acc;
    for (int32_t i = 0; i < n; i++) {
        const int phys = lg_physical_device(logical_devs[i]);
        bool seen = false;
        for (const auto& a : acc) seen = seen || a.location.id == phys;
        if (seen) continue;
        hipMemAccessDesc d = {};
        d.location.type = hipMemLocationTypeDevice;
        d.location.id = phys;
        d.flags = hipMemAccessFlagsProtReadWrite;
        acc.push_back(d);
    }
    if (acc.empty()) return 1;
    const hipError_t e = hipMemSetAccess(ptr, bytes, acc.data(), acc.size());
    if (e != hipSuccess) {          // (the caller decides: a bulk leg without peer access to the arenas cannot run, everything else can)
        (void)hipGetLastError();
        printf("legion_hip: hipMemSetAccess for %d device(s) failed: '%s'\n", (int)acc.size(), hipGetErrorString(e));
        return 0;
    }
    return 1;
}

// Other processes: a detached thread hands the chunks' file descriptors (64 per SCM_RIGHTS message, in mapping order) to whoever
// connects to the abstract unix socket `name` -- if it is a process of the same user (SO_PEERCRED: a descriptor to device memory is a
// capability, and the abstract namespace has no file permissions).  `ptr` must be an EXPORTABLE scattered allocation.  The thread and
// its socket go when the allocation is freed (d_free_scattered shuts the socket down; the thread, woken by that, closes it).
extern "C" int32_t lg_scattered_serve(void* ptr, const char* name)
{
    int32_t n_chunks = 0;
    if (lg_scattered_info(ptr, &n_chunks) <= 0) return 0;
    if (!lg_scattered_exportable(ptr)) {
        printf("legion_hip: %s: the allocation was not created exportable (d_alloc_scattered_exportable): its chunks cannot be served\n", name);
        return 0;
    }
    const int ls = socket(AF_UNIX, SOCK_STREAM | SOCK_CLOEXEC, 0);
    if (ls < 0) return 0;
    sockaddr_un addr;
    memset(&addr, 0, sizeof(addr));
    addr.sun_family = AF_UNIX;
    const int len = snprintf(addr.sun_path + 1, sizeof(addr.sun_path) - 1, "%s", name);
    if (bind(ls, (sockaddr*)&addr, (socklen_t)(offsetof(sockaddr_un, sun_path) + 1 + len)) != 0 || listen(ls, 16) != 0) { close(ls); return 0; }
    {
        std::lock_guard<std::mutex> lk(g_scatter_mu);
        auto it = g_scatter_live.find(ptr);
        if (it == g_scatter_live.end()) { close(ls); return 0; }
        it->second.listeners.push_back(ls);
    }
    int hip_dev = 0;
    (void)hipGetDevice(&hip_dev);                      // (the caller's device: the allocation's)
    std::thread([ls, ptr, n_chunks, hip_dev]() {
        (void)hipSetDevice(hip_dev);
        for (;;) {
            const int c = accept4(ls, nullptr, nullptr, SOCK_CLOEXEC);
            if (c < 0) {
                // the allocation has been freed (d_free_scattered took it out of the registry, then shut the socket down): close and go.
                // Anything else -- EINTR, ECONNABORTED, EMFILE / ENFILE, ENOMEM ... -- is transient: a later peer must still be served.
                bool freed;
                {
                    std::lock_guard<std::mutex> lk(g_scatter_mu);
                    auto it = g_scatter_live.find(ptr);
                    freed = it == g_scatter_live.end() || std::find(it->second.listeners.begin(), it->second.listeners.end(), ls) == it->second.listeners.end();
                }
                if (freed) { close(ls); return; }
                if (errno != EINTR) usleep(2000);
                continue;
            }
            {   // a peer that connects and never reads must not block the only serving thread for ever
                timeval tv = {2, 0};
                (void)setsockopt(c, SOL_SOCKET, SO_SNDTIMEO, &tv, sizeof(tv));
            }
            {
                ucred cr;
                socklen_t cl = sizeof(cr);
                if (getsockopt(c, SOL_SOCKET, SO_PEERCRED, &cr, &cl) != 0 || cr.uid != geteuid()) {
                    printf("legion_hip: refused a request for device-memory descriptors from uid %d (pid %d)\n", (int)cr.uid, (int)cr.pid);
                    fflush(stdout);
                    close(c);
                    continue;
                }
            }
            bool ok = true;
            for (int32_t i0 = 0; i0 < n_chunks && ok; i0 += 64) {
                const int32_t n = std::min(64, n_chunks - i0);
                int fds[64];
                for (int32_t i = 0; i < n; i++) { fds[i] = lg_scattered_export_fd(ptr, i0 + i); ok = ok && fds[i] >= 0; }
                if (ok) {
                    char payload = 'f';
                    iovec io = {&payload, 1};
                    alignas(cmsghdr) char ctl[CMSG_SPACE(sizeof(int) * 64)];
                    memset(ctl, 0, sizeof(ctl));
                    msghdr msg;
                    memset(&msg, 0, sizeof(msg));
                    msg.msg_iov = &io; msg.msg_iovlen = 1; msg.msg_control = ctl; msg.msg_controllen = CMSG_SPACE(sizeof(int) * n);
                    cmsghdr* cm = CMSG_FIRSTHDR(&msg);
                    cm->cmsg_level = SOL_SOCKET; cm->cmsg_type = SCM_RIGHTS; cm->cmsg_len = CMSG_LEN(sizeof(int) * n);
                    memcpy(CMSG_DATA(cm), fds, sizeof(int) * n);
                    ok = sendmsg(c, &msg, MSG_NOSIGNAL) == 1;
                }
                for (int32_t i = 0; i < n; i++) if (fds[i] >= 0) close(fds[i]);
            }
            close(c);
        }
    }).detach();
    return 1;
}

// ... and the other side: connect to `name`, receive n_chunks descriptors, import and map them back to back, accessible to THIS
// process's current device.  Returns the base address or null (why: printed).  lg_scattered_unmap_remote gives everything back.
struct RemoteMap { size_t chunk_bytes; int32_t n_chunks; };
static std::map<void*, RemoteMap> g_remote_maps;
extern "C" void* lg_scattered_map_remote(const char* name, int32_t n_chunks, int64_t chunk_bytes)
{
    if (n_chunks <= 0 || chunk_bytes <= 0) return nullptr;
    const int c = socket(AF_UNIX, SOCK_STREAM | SOCK_CLOEXEC, 0);
    if (c < 0) return nullptr;
    sockaddr_un addr;
    memset(&addr, 0, sizeof(addr));
    addr.sun_family = AF_UNIX;
    const int len = snprintf(addr.sun_path + 1, sizeof(addr.sun_path) - 1, "%s", name);
    if (connect(c, (sockaddr*)&addr, (socklen_t)(offsetof(sockaddr_un, sun_path) + 1 + len)) != 0) { close(c); printf("legion_hip: connect(%s) failed\n", name); return nullptr; }
    {
        ucred cr;
        socklen_t cl = sizeof(cr);
        if (getsockopt(c, SOL_SOCKET, SO_PEERCRED, &cr, &cl) != 0 || cr.uid != geteuid()) { close(c); printf("legion_hip: %s belongs to another user\n", name); return nullptr; }
    }
    int dev = 0;
    HIP_CALL(hipGetDevice(&dev));
    void* base = nullptr;
    HIP_CALL(hipMemAddressReserve(&base, (size_t)n_chunks * (size_t)chunk_bytes, 0, nullptr, 0));
    int got = 0;
    bool ok = true;
    while (got < n_chunks && ok) {
        char payload = 0;
        iovec io = {&payload, 1};
        alignas(cmsghdr) char ctl[CMSG_SPACE(sizeof(int) * 64)];
        msghdr msg;
        memset(&msg, 0, sizeof(msg));
        msg.msg_iov = &io; msg.msg_iovlen = 1; msg.msg_control = ctl; msg.msg_controllen = sizeof(ctl);
        if (recvmsg(c, &msg, MSG_CMSG_CLOEXEC) != 1) { ok = false; break; }
        cmsghdr* cm = CMSG_FIRSTHDR(&msg);
        if (cm == nullptr || cm->cmsg_level != SOL_SOCKET || cm->cmsg_type != SCM_RIGHTS) { ok = false; break; }
        const int n = (int)((cm->cmsg_len - CMSG_LEN(0)) / sizeof(int));
        int fds[64];
        memcpy(fds, CMSG_DATA(cm), sizeof(int) * (size_t)n);
        for (int i = 0; i < n; i++) {
            if (ok && got < n_chunks) {
                hipMemGenericAllocationHandle_t h;
                if (vmm_import_fd(&h, fds[i]) != hipSuccess) {
                    (void)hipGetLastError();
                    ok = false;
                } else {
                    if (hipMemMap((char*)base + (size_t)got * (size_t)chunk_bytes, (size_t)chunk_bytes, 0, h, 0) != hipSuccess) { (void)hipGetLastError(); ok = false; }
                    else got++;
                    if (hipMemRelease(h) != hipSuccess) (void)hipGetLastError();
                }
            }
            close(fds[i]);
        }
    }
    close(c);
    if (ok) {
        hipMemAccessDesc acc = {};
        acc.location.type = hipMemLocationTypeDevice;
        acc.location.id = dev;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        if (hipMemSetAccess(base, (size_t)n_chunks * (size_t)chunk_bytes, &acc, 1) != hipSuccess) { (void)hipGetLastError(); ok = false; }
    }
    if (!ok) {
        for (int i = 0; i < got; i++) (void)hipMemUnmap((char*)base + (size_t)i * (size_t)chunk_bytes, (size_t)chunk_bytes);
        (void)hipMemAddressFree(base, (size_t)n_chunks * (size_t)chunk_bytes);
        (void)hipGetLastError();
        printf("legion_hip: mapping %d chunks of %s failed\n", n_chunks, name);
        return nullptr;
    }
    std::lock_guard<std::mutex> lk(g_scatter_mu);
    g_remote_maps[base] = RemoteMap{(size_t)chunk_bytes, n_chunks};
    return base;
}
extern "C" void lg_scattered_unmap_remote(void* base)
{
    RemoteMap m;
    {
        std::lock_guard<std::mutex> lk(g_scatter_mu);
        auto it = g_remote_maps.find(base);
        if (it == g_remote_maps.end()) return;
        m = it->second;
        g_remote_maps.erase(it);
    }
    for (int32_t i = 0; i < m.n_chunks; i++)
        if (hipMemUnmap((char*)base + (size_t)i * m.chunk_bytes, m.chunk_bytes) != hipSuccess) (void)hipGetLastError();
    if (hipMemAddressFree(base, (size_t)m.n_chunks * m.chunk_bytes) != hipSuccess) (void)hipGetLastError();
}

static bool d_free_scattered(void* ptr)
{
    ScatterLive live;
    {
        std::lock_guard<std::mutex> lk(g_scatter_mu);
        auto it = g_scatter_live.find(ptr);
        if (it == g_scatter_live.end()) return false;
        live = std::move(it->second);
        g_scatter_live.erase(it);
    }
    for (int ls : live.listeners) shutdown(ls, SHUT_RDWR);      // wakes the serving thread out of accept(); it finds the allocation gone and closes the socket itself
    HIP_CALL(hipMemUnmap(ptr, live.bytes));
    HIP_CALL(hipMemAddressFree(ptr, live.bytes));
    for (auto h : live.chunks) HIP_CALL(hipMemRelease(h));
    return true;
}

// ---- a pipeline's lane-PRIVATE arrays (slot arrays, claim lists, frontier headers ...) from one block of shuffled chunks as well: while a
//      thread has a private arena set, its d_alloc_space calls are carved from it (never freed one by one: the block goes when the pipeline
//      goes).  Measured: +0.9 % on the whole job, +4 % on the sampler chain, against both separate allocations and one plain block -----------
namespace {
struct PrivArena { char* base; int64_t bytes, used; };
thread_local PrivArena* g_priv = nullptr;
thread_local int64_t g_count_bytes = -1;          // >= 0: d_alloc_space adds up what this thread asks for (lg_alloc_count_begin / _end)
std::mutex g_priv_mu;
std::vector<std::pair<char*, int64_t>> g_priv_ranges;
}
extern "C" void* lg_private_arena_begin(int64_t bytes, int32_t scatter_mb)
{
    PrivArena* a = new PrivArena();
    a->base = (char*)(scatter_mb > 0 ? d_alloc_scattered(bytes, scatter_mb) : nullptr);
    if (a->base == nullptr) HIP_CALL(hipMalloc((void**)&a->base, (size_t)bytes));
    a->bytes = bytes;
    a->used = 0;
    { std::lock_guard<std::mutex> lk(g_priv_mu); g_priv_ranges.push_back({a->base, bytes}); }
    g_priv = a;
    return a;
}
extern "C" void lg_private_arena_end() { g_priv = nullptr; }
extern "C" void lg_alloc_count_begin() { g_count_bytes = 0; }
extern "C" int64_t lg_alloc_count_end() { const int64_t v = g_count_bytes; g_count_bytes = -1; return v; }
extern "C" void lg_private_arena_free(void* handle)
{
    PrivArena* a = (PrivArena*)handle;
    if (!a) return;
    {
        std::lock_guard<std::mutex> lk(g_priv_mu);
        for (size_t i = 0; i < g_priv_ranges.size(); i++)
            if (g_priv_ranges[i].first == a->base) { g_priv_ranges.erase(g_priv_ranges.begin() + (long)i); break; }
    }
    if (!d_free_scattered(a->base)) HIP_CALL(hipFree(a->base));
    delete a;
}
static bool in_private_arena(void* p)
{
    std::lock_guard<std::mutex> lk(g_priv_mu);
    for (auto& r : g_priv_ranges) if ((char*)p >= r.first && (char*)p < r.first + r.second) return true;
    return false;
}

extern "C" void* d_alloc_space(int64_t num_bytes)
{
    if (g_count_bytes >= 0) g_count_bytes += ((num_bytes > 0 ? num_bytes : 16) + 255) & ~(int64_t)255;
    if (g_priv != nullptr) {
        const int64_t need = ((num_bytes > 0 ? num_bytes : 16) + 255) & ~(int64_t)255;
        if (g_priv->used + need <= g_priv->bytes) { void* p = g_priv->base + g_priv->used; g_priv->used += need; return p; }
    }
    void* ret = nullptr;
    HIP_CALL(hipMalloc(&ret, num_bytes > 0 ? (size_t)num_bytes : 16));
    return ret;
}

static bool lg_ipc_try_export(void* handle64, void* dev_ptr, int attempts, hipError_t* last)
{
    for (int attempt = 0; attempt < attempts; attempt++) {
        *last = hipIpcGetMemHandle((hipIpcMemHandle_t*)handle64, dev_ptr);
        if (*last == hipSuccess) return true;
        (void)hipGetLastError();
        usleep(25000);
    }
    return false;
}

static void lg_ipc_diagnose(void* dev_ptr, hipError_t e, const char* file, int line)
{
    hipPointerAttribute_t at;
    memset(&at, 0, sizeof(at));
    const hipError_t pe = hipPointerGetAttributes(&at, dev_ptr);
    (void)hipGetLastError();
    const char* legacy = getenv("HSA_ENABLE_IPC_MODE_LEGACY");
    printf("HIP failure %s:%d: '%s' (hipIpcGetMemHandle of %p; attributes %s: device %d, type %d, base %p; "
           "HSA_ENABLE_IPC_MODE_LEGACY=%s)\n", file, line, hipGetErrorString(e), dev_ptr, hipGetErrorString(pe), at.device,
           (int)at.type, at.devicePointer, legacy ? legacy : "(unset)");
    fflush(stdout);
}

void lg_ipc_export(void* handle64, void* dev_ptr, const char* file, int line)
{
    hipError_t e = hipSuccess;
    if (lg_ipc_try_export(handle64, dev_ptr, 20, &e)) return;
    lg_ipc_diagnose(dev_ptr, e, file, line);
    exit(EXIT_FAILURE);
}

// A fresh device buffer together with its IPC handle.  About one server start in a hundred on this pool (ROCm 7.2, dmabuf
// IPC) finds EVERY small block of the process unexportable ('invalid argument', pointer attributes fine, mode variable
// set) while other starts export the same sizes without trouble -- small allocations are fragments of a shared runtime
// block there.  A block that keeps failing is set aside (not freed: the allocator would hand it out again) and a block of
// its own (>= 2 MiB, not a fragment) is tried instead.
void* lg_alloc_exported(int64_t num_bytes, void* handle64, const char* file, int line)
{
    hipError_t e = hipSuccess;
    void* p = nullptr;
    for (int block = 0; block < 4; block++) {
        const int64_t bytes = block == 0 ? num_bytes : std::max<int64_t>(num_bytes, ((int64_t)2 << 20) * block + 4096);
        p = d_alloc_space(bytes);
        if (lg_ipc_try_export(handle64, p, block == 0 ? 3 : 6, &e)) {
            if (block > 0) { printf("legion_hip: IPC export succeeded with a block of %lld bytes\n", (long long)bytes); fflush(stdout); }
            return p;
        }
        lg_ipc_diagnose(p, e, file, line);
    }
    exit(EXIT_FAILURE);
}

extern "C" void d_free_space(void* d_ptr)
{
    if (d_ptr && !in_private_arena(d_ptr) && !d_free_scattered(d_ptr)) HIP_CALL(hipFree(d_ptr));
}

extern "C" void* host_alloc_space(int64_t num_bytes)
{
    void* host_ptr = nullptr;
    void* ret = nullptr;
    HIP_CALL(hipHostMalloc(&host_ptr, num_bytes > 0 ? (size_t)num_bytes : 16, hipHostMallocMapped));
    HIP_CALL(hipHostGetDevicePointer(&ret, host_ptr, 0));
    return ret;
}

// spill-over tier: mapped pinned host memory the GPU dereferences over PCIe (SS/engine/server_imp.cuh:41-51,
// storage_management.cu:106-107,161); returns the device-side pointer, *host_ptr_out the host-side one
extern "C" void* legion_host_alloc(int64_t num_bytes, void** host_ptr_out)
{
    void* host_ptr = nullptr;
    void* dev_ptr = nullptr;
    HIP_CALL(hipHostMalloc(&host_ptr, num_bytes > 0 ? (size_t)num_bytes : 16, hipHostMallocMapped));
    HIP_CALL(hipHostGetDevicePointer(&dev_ptr, host_ptr, 0));
    if (host_ptr_out) *host_ptr_out = host_ptr;
    return dev_ptr;
}

extern "C" void legion_host_free(void* host_ptr)
{
    if (host_ptr) HIP_CALL(hipHostFree(host_ptr));
}

extern "C" int32_t legion_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" const char* legion_version(void) { return "legion-hip 0.1 (gfx950)"; }

// =============================================================================================
class CompleteGraphStorage : public GraphStorage {
public:
    void Build(BuildInfo* info) override
    {
        partition_count_ = info->partition_count;
        node_num_ = info->total_num_nodes;
        edge_num_ = info->total_edge_num;
        csr_node_index_cpu_ = info->csr_node_index;
        csr_dst_node_ids_cpu_ = info->csr_dst_node_ids;
        csr_node_index_.assign(partition_count_, nullptr);
        csr_dst_node_ids_.assign(partition_count_, nullptr);
        h_index_tab_.assign(partition_count_, std::vector<int64_t*>(partition_count_ + 1, nullptr));
        h_dst_tab_.assign(partition_count_, std::vector<int32_t*>(partition_count_ + 1, nullptr));
        row_hdr_.assign(partition_count_, nullptr);
        topo_index_.assign(partition_count_, nullptr);
        topo_col_.assign(partition_count_, nullptr);
        csr_dst_x_.assign(partition_count_, nullptr);
        colx_full_.assign(partition_count_, nullptr);
        for (int32_t i = 0; i < partition_count_; i++) {
            if (!lg_is_local(i)) continue;
            SetGPUDevice(i);
            csr_node_index_[i] = (int64_t**)d_alloc_space((partition_count_ + 1) * sizeof(int64_t*));
            csr_dst_node_ids_[i] = (int32_t**)d_alloc_space((partition_count_ + 1) * sizeof(int32_t*));
            // every GPU can reach the full CSR through slot P (the reference initialises only
            // GPU 0's table before GraphCache copies it around, graph_storage.cu:62,77-80)
            h_index_tab_[i][partition_count_] = csr_node_index_cpu_;
            h_dst_tab_[i][partition_count_] = csr_dst_node_ids_cpu_;
            Upload(i);
            // per-vertex row headers: everything starts in the full CSR (slot P)
            row_hdr_[i] = (RowHdr*)d_alloc_space((int64_t)node_num_ * sizeof(RowHdr));
            lg::init_row_headers(nullptr, row_hdr_[i], csr_node_index_cpu_, node_num_, partition_count_);
            HIP_CALL(hipDeviceSynchronize());
        }
    }

    // SS/storage/graph_storage.cu:76-111: GPU i of clique Ki caches vertices QT[row*Kg + i]
    void GraphCache(int32_t* QT, int32_t Ki, int32_t Kg, int32_t capacity) override
    {
        GraphCacheBuildLocal(QT, Ki, Kg, capacity);
        GraphCacheLink(QT, Ki, Kg, capacity);
    }

    void GraphCacheBuildLocal(int32_t* QT, int32_t Ki, int32_t Kg, int32_t capacity) override
    {
        for (int32_t i = 0; i < Kg; i++) {
            const int32_t dev = Ki * Kg + i;
            if (!lg_is_local(dev)) continue;
            SetGPUDevice(dev);
            // a new fill: headers that an earlier fill pointed into ITS cached CSR go back to the full CSR first (a vertex
            // cached then and not now kept a row offset into arrays the pointer table no longer names)
            lg::init_row_headers(nullptr, row_hdr_[dev], csr_node_index_cpu_, node_num_, partition_count_);
            HIP_CALL(hipDeviceSynchronize());
            int64_t* neighbor_count = (int64_t*)d_alloc_space((int64_t)capacity * sizeof(int64_t));
            lg::topo_neighbor_count(nullptr, QT, Kg, i, capacity, node_num_, csr_node_index_cpu_, neighbor_count);
            int64_t* d_csr_node_index = (int64_t*)d_alloc_space(((int64_t)capacity + 1) * sizeof(int64_t));
            HIP_CALL(hipMemset(d_csr_node_index, 0, ((size_t)capacity + 1) * sizeof(int64_t)));
            if (capacity > 0) lg::inclusive_scan_i64(nullptr, neighbor_count, d_csr_node_index + 1, capacity);
            int64_t cached_edges = 0;
            HIP_CALL(hipMemcpy(&cached_edges, d_csr_node_index + capacity, sizeof(int64_t), hipMemcpyDeviceToHost));
            int32_t* d_csr_dst_node_ids = (int32_t*)d_alloc_space(cached_edges * sizeof(int32_t));
            lg::topo_fill_up(nullptr, QT, Kg, i, capacity, node_num_, csr_node_index_cpu_,
                             csr_dst_node_ids_cpu_, d_csr_node_index, d_csr_dst_node_ids);
            HIP_CALL(hipDeviceSynchronize());
            d_free_space(neighbor_count);
            owned_.push_back(d_csr_node_index);
            owned_.push_back(d_csr_dst_node_ids);
            topo_index_[dev] = d_csr_node_index;
            topo_col_[dev] = d_csr_dst_node_ids;
        }
    }

    void GraphCacheLink(int32_t* QT, int32_t Ki, int32_t Kg, int32_t capacity) override
    {
        for (int32_t j = 0; j < Kg; j++) {                // every LOCAL member of the clique ...
            const int32_t member = Ki * Kg + j;
            if (!lg_is_local(member)) continue;
            SetGPUDevice(member);
            for (int32_t i = 0; i < Kg; i++) {            // ... sees every member's cached CSR it knows of
                const int32_t dev = Ki * Kg + i;
                if (topo_index_[dev] == nullptr) continue;
                h_index_tab_[member][dev] = topo_index_[dev];
                h_dst_tab_[member][dev] = topo_col_[dev];
                // resolve the vertices that member caches to its CSR (reads its indptr: a peer load when
                // the member is another GPU)
                lg::cache_row_headers(nullptr, row_hdr_[member], QT, Kg, i, capacity, node_num_, topo_index_[dev], dev);
                HIP_CALL(hipDeviceSynchronize());
            }
            Upload(member);
        }
    }

    void SetPeerCSR(int32_t dev, int64_t* csr_node_index, int32_t* csr_dst_node_ids) override
    {
        topo_index_[dev] = csr_node_index;
        topo_col_[dev] = csr_dst_node_ids;
    }
    int64_t* CachedCSRIndex(int32_t dev) const override { return topo_index_[dev]; }
    int32_t* CachedCSRDst(int32_t dev) const override { return topo_col_[dev]; }

    // column slots (legion_core.h): {neighbour id, node_map[neighbour]} pairs of the FULL column array, for GPU dev
    void BuildColumnSlots(int32_t dev, const int32_t* node_map, uint64_t stamp) override
    {
        if (dev < 0 || dev >= partition_count_ || !lg_is_local(dev)) return;
        DropColumnSlots(dev);
        if ((int32_t)colx_stamp_.size() < partition_count_) colx_stamp_.assign(partition_count_, 0);
        const int32_t mode = lg::tuning().col_slots;
        if (mode == 0 || node_map == nullptr || edge_num_ <= 0) return;
        SetGPUDevice(dev);
        const int64_t bytes = edge_num_ * 8;
        if (mode < 0) {
            hipPointerAttribute_t at;
            memset(&at, 0, sizeof(at));
            if (hipPointerGetAttributes(&at, csr_dst_node_ids_cpu_) != hipSuccess || at.type != hipMemoryTypeDevice) {
                (void)hipGetLastError();
                return;                       // column array in (pinned) host memory: the spill-over configuration keeps HBM free
            }
            // What is free NOW is what the full tables, the caches and the maps left (FillUp has run); the lanes of the launch
            // groups come afterwards and size themselves to what remains (bench.py, GPURunner::CreateGroups).  The copy is worth
            // 3-7 % of the whole job (the gather stops fetching a 128-byte line of node_map per row), so it is taken whenever it
            // fits half of the free HBM and leaves 24 GB for the lanes -- uk-union's 44 GB copy beside 160 GB of tables on a
            // 288 GB part (round 3 declined it: a flat quarter of the free memory)
            size_t free_b = 0, total_b = 0;
            HIP_CALL(hipMemGetInfo(&free_b, &total_b));
            if (bytes > (int64_t)(free_b / 2) || (int64_t)free_b - bytes < ((int64_t)24 << 30)) return;
        }
        colx_full_[dev] = (int32_t*)d_alloc_space(bytes);
        lg::build_column_slots(nullptr, csr_dst_node_ids_cpu_, node_map, colx_full_[dev], edge_num_);
        std::vector<int32_t*> tab(partition_count_ + 1, nullptr);
        tab[partition_count_] = colx_full_[dev];
        csr_dst_x_[dev] = (int32_t**)d_alloc_space((partition_count_ + 1) * sizeof(int32_t*));
        HIP_CALL(hipMemcpy(csr_dst_x_[dev], tab.data(), tab.size() * sizeof(int32_t*), hipMemcpyHostToDevice));
        HIP_CALL(hipDeviceSynchronize());
        colx_stamp_[dev] = stamp;
    }
    uint64_t ColumnSlotsStamp(int32_t dev) const override
    {
        return (dev >= 0 && dev < (int32_t)colx_stamp_.size() && dev < (int32_t)csr_dst_x_.size() && csr_dst_x_[dev] != nullptr) ? colx_stamp_[dev] : 0;
    }
    void DropColumnSlots(int32_t dev) override
    {
        if (dev < 0 || dev >= (int32_t)csr_dst_x_.size() || csr_dst_x_[dev] == nullptr) return;
        SetGPUDevice(dev);
        HIP_CALL(hipDeviceSynchronize());
        d_free_space(csr_dst_x_[dev]);
        d_free_space(colx_full_[dev]);
        csr_dst_x_[dev] = nullptr;
        colx_full_[dev] = nullptr;
        if (dev < (int32_t)colx_stamp_.size()) colx_stamp_[dev] = 0;
    }
    int32_t** GetCSRXMatrix(int32_t part_id) const override { return csr_dst_x_.empty() ? nullptr : csr_dst_x_[part_id]; }
    const int32_t* GetColumnSlotsFull(int32_t part_id) const override { return colx_full_.empty() ? nullptr : colx_full_[part_id]; }

    void Finalize() override
    {
        for (int32_t i = 0; i < (int32_t)csr_dst_x_.size(); i++) DropColumnSlots(i);
        for (void* p : owned_) d_free_space(p);
        owned_.clear();
        for (RowHdr* p : row_hdr_) d_free_space(p);
        row_hdr_.clear();
        for (int32_t i = 0; i < partition_count_; i++) {
            if (!lg_is_local(i)) continue;
            d_free_space(csr_node_index_[i]);
            d_free_space(csr_dst_node_ids_[i]);
        }
        csr_node_index_.clear();
        csr_dst_node_ids_.clear();
    }

    int32_t GetPartitionCount() const override { return partition_count_; }
    int64_t** GetCSRNodeIndex(int32_t part_id) const override { return csr_node_index_[part_id]; }
    int32_t** GetCSRNodeMatrix(int32_t part_id) const override { return csr_dst_node_ids_[part_id]; }
    int64_t* GetCSRNodeIndexCPU() const override { return csr_node_index_cpu_; }
    int32_t* GetCSRNodeMatrixCPU() const override { return csr_dst_node_ids_cpu_; }
    int32_t NodeNum() const override { return node_num_; }
    int64_t EdgeNum() const override { return edge_num_; }
    const RowHdr* GetRowHeaders(int32_t part_id) const override { return row_hdr_[part_id]; }

private:
    void Upload(int32_t dev)
    {
        SetGPUDevice(dev);
        HIP_CALL(hipMemcpy(csr_node_index_[dev], h_index_tab_[dev].data(),
                           (partition_count_ + 1) * sizeof(int64_t*), hipMemcpyHostToDevice));
        HIP_CALL(hipMemcpy(csr_dst_node_ids_[dev], h_dst_tab_[dev].data(),
                           (partition_count_ + 1) * sizeof(int32_t*), hipMemcpyHostToDevice));
    }

    int32_t partition_count_ = 0;
    int32_t node_num_ = 0;
    int64_t edge_num_ = 0;
    std::vector<int64_t**> csr_node_index_;
    std::vector<int32_t**> csr_dst_node_ids_;
    std::vector<std::vector<int64_t*>> h_index_tab_;
    std::vector<std::vector<int32_t*>> h_dst_tab_;
    int64_t* csr_node_index_cpu_ = nullptr;
    int32_t* csr_dst_node_ids_cpu_ = nullptr;
    std::vector<void*> owned_;
    std::vector<RowHdr*> row_hdr_;
    std::vector<int64_t*> topo_index_;   // [P] cached CSR of each GPU (local build or peer pointer)
    std::vector<int32_t*> topo_col_;
    std::vector<int32_t**> csr_dst_x_;   // [P] device tables of pair arrays (column slots), null until built
    std::vector<int32_t*> colx_full_;    // [P] this GPU's {id, feature-cache slot} copy of the full column array
    std::vector<uint64_t> colx_stamp_;   // [P] (cache uid, fill generation) the pairs were built from
};

extern "C" GraphStorage* NewCompleteGraphStorage() { return new CompleteGraphStorage(); }

// =============================================================================================
class CompleteFeatureStorage : public FeatureStorage {
public:
    void Build(BuildInfo* info, int /*in_memory_mode*/) override
    {
        Configure(info->partition_count, info->total_num_nodes, info->float_feature_len,
                  info->host_float_feature);
        for (int32_t p = 0; p < partition_count_; p++) {
            if (p < (int32_t)info->training_set_ids.size())
                SetIds(p, TRAINMODE, info->training_set_ids[p].data(), info->training_labels[p].data(),
                       (int32_t)info->training_set_ids[p].size());
            if (p < (int32_t)info->validation_set_ids.size())
                SetIds(p, VALIDMODE, info->validation_set_ids[p].data(), info->validation_labels[p].data(),
                       (int32_t)info->validation_set_ids[p].size());
            if (p < (int32_t)info->testing_set_ids.size())
                SetIds(p, TESTMODE, info->testing_set_ids[p].data(), info->testing_labels[p].data(),
                       (int32_t)info->testing_set_ids[p].size());
        }
    }

    void Configure(int32_t partition_count, int32_t total_num_nodes, int32_t float_feature_len,
                   float* all_float_feature)
    {
        partition_count_ = partition_count;
        total_num_nodes_ = total_num_nodes;
        float_feature_len_ = float_feature_len;
        float_feature_ = all_float_feature;
        for (int m = 0; m < 3; m++) {
            ids_[m].assign(partition_count, nullptr);
            labels_[m].assign(partition_count, nullptr);
            size_[m].assign(partition_count, 0);
        }
    }

    void SetIds(int32_t dev_id, int32_t mode, const int32_t* host_ids, const int32_t* host_labels,
                int32_t count) override
    {
        SetGPUDevice(dev_id);
        d_free_space(ids_[mode][dev_id]);
        d_free_space(labels_[mode][dev_id]);
        ids_[mode][dev_id] = (int32_t*)d_alloc_space((int64_t)count * sizeof(int32_t));
        labels_[mode][dev_id] = (int32_t*)d_alloc_space((int64_t)count * sizeof(int32_t));
        if (count > 0) {
            HIP_CALL(hipMemcpy(ids_[mode][dev_id], host_ids, (size_t)count * 4, hipMemcpyHostToDevice));
            if (host_labels)
                HIP_CALL(hipMemcpy(labels_[mode][dev_id], host_labels, (size_t)count * 4, hipMemcpyHostToDevice))
            else
                HIP_CALL(hipMemset(labels_[mode][dev_id], 0, (size_t)count * 4));   // v2: labels are all 0 (F7)
        }
        size_[mode][dev_id] = count;
    }

    void Finalize() override
    {
        for (int m = 0; m < 3; m++)
            for (size_t p = 0; p < ids_[m].size(); p++) {
                d_free_space(ids_[m][p]);
                d_free_space(labels_[m][p]);
                ids_[m][p] = labels_[m][p] = nullptr;
            }
    }

    int32_t* GetTrainingSetIds(int32_t p) const override { return ids_[TRAINMODE][p]; }
    int32_t* GetValidationSetIds(int32_t p) const override { return ids_[VALIDMODE][p]; }
    int32_t* GetTestingSetIds(int32_t p) const override { return ids_[TESTMODE][p]; }
    int32_t* GetTrainingLabels(int32_t p) const override { return labels_[TRAINMODE][p]; }
    int32_t* GetValidationLabels(int32_t p) const override { return labels_[VALIDMODE][p]; }
    int32_t* GetTestingLabels(int32_t p) const override { return labels_[TESTMODE][p]; }
    int32_t TrainingSetSize(int32_t p) const override { return size_[TRAINMODE][p]; }
    int32_t ValidationSetSize(int32_t p) const override { return size_[VALIDMODE][p]; }
    int32_t TestingSetSize(int32_t p) const override { return size_[TESTMODE][p]; }
    int32_t TotalNodeNum() const override { return total_num_nodes_; }
    float* GetAllFloatFeature() const override { return float_feature_; }
    int32_t GetFloatFeatureLen() const override { return float_feature_len_; }

private:
    int32_t partition_count_ = 0, total_num_nodes_ = 0, float_feature_len_ = 0;
    float* float_feature_ = nullptr;
    std::vector<int32_t*> ids_[3], labels_[3];
    std::vector<int32_t> size_[3];
};

extern "C" FeatureStorage* NewCompleteFeatureStorage() { return new CompleteFeatureStorage(); }

// =============================================================================================
LanePtrs MemoryPool::HostLane(int32_t pipe) const
{
    LanePtrs h;
    memset(&h, 0, sizeof(h));
    h.sampled_ids = sampled_ids_[pipe];
    h.labels = labels_[pipe];
    h.agg_src_ids = agg_src_ids_;
    h.agg_dst_ids = agg_dst_ids_;
    h.agg_src_off = agg_src_off_[pipe];
    h.agg_dst_off = agg_dst_off_[pipe];
    h.tmp_part_ind = tmp_part_ind_;
    h.err_flag = err_dev;
    h.counter_mirror = counter_mirror_dev;
    h.claim_pairs = claim_pairs;
    h.run_off = run_off;
    h.claim_cnt = claim_cnt;
    h.claim_cap = claim_cap;
    h.ids_cap = num_ids;
    h.lds_buckets = 1 << lds_bucket_bits;
    h.known_pairs = known_pairs;
    h.known_cnt = known_cnt;
    h.known_cap = known_cap;
    h.node_counter = node_counter_[pipe];
    h.edge_counter = edge_counter_[pipe];
    h.slot_dst = slot_dst;
    h.slot_pos = slot_pos;
    h.slot_fs = slot_fs;
    h.node_slot = node_slot;
    h.max_slots = max_slots;
    h.tile_state = tile_state;
    h.hop_scratch = hop_scratch;
    h.fh_edge = fh_edge;
    h.cache_search_buffer = cache_search_buffer_;
    h.float_features = float_features_[pipe];
    int64_t rows = feature_rows < num_ids ? feature_rows : num_ids;
    h.feature_rows = (int32_t)rows;
    return h;
}

const LanePtrs* MemoryPool::DeviceLane()
{
    if (lanes_dirty_ || d_lanes_ == nullptr || uploaded_rows_ != feature_rows) {
        SetGPUDevice(dev_id);
        if (d_lanes_ == nullptr) d_lanes_ = (LanePtrs*)d_alloc_space((int64_t)pipeline_depth_ * sizeof(LanePtrs));
        std::vector<LanePtrs> h(pipeline_depth_);
        for (int32_t i = 0; i < pipeline_depth_; i++) h[i] = HostLane(i);
        HIP_CALL(hipMemcpy(d_lanes_, h.data(), h.size() * sizeof(LanePtrs), hipMemcpyHostToDevice));
        lanes_dirty_ = false;
        uploaded_rows_ = feature_rows;
    }
    return d_lanes_ + current_pipe_;
}

void MemoryPool::Finalize()
{
    SetGPUDevice(dev_id);
    d_free_space(d_lanes_);
    d_lanes_ = nullptr;
    d_free_space(cache_search_buffer_);
    d_free_space(claim_pairs);
    d_free_space(run_off);
    d_free_space(claim_cnt);
    claim_cnt = nullptr;
    d_free_space(known_pairs);
    d_free_space(known_cnt);
    known_pairs = nullptr;
    known_cnt = nullptr;
    claim_pairs = nullptr;
    run_off = nullptr;
    if (err_host) HIP_CALL(hipHostFree(err_host));
    err_host = err_dev = nullptr;
    d_free_space(agg_src_ids_);
    d_free_space(agg_dst_ids_);
    d_free_space(tmp_part_ind_);
    d_free_space(tmp_part_off_);
    d_free_space(slot_dst);
    d_free_space(slot_pos);
    slot_pos = nullptr;
    d_free_space(slot_fs);
    d_free_space(node_slot);
    slot_fs = node_slot = nullptr;
    d_free_space(tile_state);
    d_free_space(hop_scratch);
    d_free_space(fh_edge);
    fh_edge = nullptr;
    cache_search_buffer_ = agg_src_ids_ = agg_dst_ids_ = tmp_part_off_ = nullptr;
    tmp_part_ind_ = nullptr;
    slot_dst = hop_scratch = nullptr;
    tile_state = nullptr;
    if (owns_buffers) {
        for (int i = 0; i < pipeline_depth_; i++) {
            d_free_space(float_features_[i]);
            d_free_space(labels_[i]);
            d_free_space(node_counter_[i]);
            d_free_space(edge_counter_[i]);
            d_free_space(sampled_ids_[i]);
            d_free_space(agg_src_off_[i]);
            d_free_space(agg_dst_off_[i]);
            float_features_[i] = nullptr;
            labels_[i] = node_counter_[i] = edge_counter_[i] = sampled_ids_[i] = nullptr;
            agg_src_off_[i] = agg_dst_off_[i] = nullptr;
        }
    }
}

static thread_local int64_t g_pool_claims_hint[2] = {0, 0};
void lg_set_pool_claims_hint(int64_t last_hop_edges, int64_t nodes_before_last_hop)
{
    g_pool_claims_hint[0] = last_hop_edges;
    g_pool_claims_hint[1] = nodes_before_last_hop;
}

// server-private scratch of one GPU: SS/engine/server.cu:216-234 plus the compaction scratch
void lg_pool_alloc_private(MemoryPool* mp, int32_t dev_id, int32_t total_num_nodes, int32_t batch_size,
                           const int32_t* fanout, int32_t hop_num, int32_t float_feature_len)
{
    SetGPUDevice(dev_id);
    lg::tuning_refresh();
    const LegionTuning& tune = lg::tuning();
    int64_t num_ids = batch_size, per = batch_size;         // server.cu:187-199
    mp->max_new.assign(1, batch_size);
    for (int i = 0; i < hop_num; i++) { per *= fanout[i]; num_ids += per; mp->max_new.push_back(per); }
    if (num_ids >= ((int64_t)1 << 31) - 16384) {          // positions and slot indices are int32, as in the reference (operator_impl.cu:208)
        printf("legion_hip: batch %d with this fan-out needs %lld slots / %lld ids; the limit is 2^31\n", batch_size, (long long)per,
               (long long)num_ids);
        exit(EXIT_FAILURE);
    }
    mp->dev_id = dev_id;
    mp->num_ids = (int32_t)num_ids;
    mp->max_slots = (int32_t)(hop_num > 0 ? per : batch_size);
    mp->total_num_nodes = total_num_nodes;
    mp->batch_size = batch_size;
    mp->float_feature_len = float_feature_len;
    mp->SetCacheSearchBuffer((int32_t*)d_alloc_space(num_ids * sizeof(int32_t)));
    {
        // first touches (legion_core.h): no per-vertex state; one claim list and one known list per hash bucket
        const int64_t slots = hop_num > 0 ? per : batch_size;
        const int64_t n_super = (slots + LG_SUPER - 1) / LG_SUPER + 1;
        // Buckets per lane.  Slots say how large a hop CAN get; PreSC (when this thread's creator passed its maxima on,
        // lg_set_pool_claims_hint) says how many claims the largest hop really has.  64 buckets serve a hop as long as a bucket's
        // claims fit the registers of its workgroup (LG_DEDUP_CLAIMS_BIG x 1024; it then runs its passes over sub-buckets from the
        // registers) -- e.g. B = 8000 with [15,10,5]: 6 M slots but ~0.9 M claims in hop 3 -- and the sampling kernel writes 64
        // lists itself; beyond that 256 buckets, whose lists a second kernel writes (place_kernel).  Without PreSC's numbers: by slots.
        const int64_t hint_claims = g_pool_claims_hint[0] * 11 / 10;      // (+10 %: buckets are not even; one that still outgrows the registers re-reads its list)
        const bool medium = hint_claims > 0 ? hint_claims <= (int64_t)64 * LG_DEDUP_CLAIMS_BIG * 1024 && slots <= ((int64_t)1 << 24)
                                            : slots <= LG_LDS_SLOTS_MEDIUM;
        mp->lds_bucket_bits = slots <= LG_LDS_SLOTS_SMALL ? LG_LDS_BITS_SMALL : (medium ? LG_LDS_BITS_MEDIUM : LG_LDS_BITS_LARGE);
        mp->last_hop_claims_hint = g_pool_claims_hint[0];
        if (mp->lds_bucket_bits == LG_LDS_BITS_SMALL) {
            // The small class has 8 or 16 buckets per lane.  Slots say how large a hop CAN get, not how many of them hold an
            // edge: on a dense graph (ogbn-products: 60 k edges per batch of 1024 where RMAT-26 has 35 k) a bucket of 8 holds
            // more vertices than one LDS table takes and every workgroup runs two passes (217 us instead of ~110 per
            // 256-lane group).  PreSC has seen the real numbers: 16 buckets where 8 would overflow one pass, else 8 (which
            // is 8 us faster per group where both fit).
            const int64_t one_pass = LG_LDS_TABLE / 16 * LG_LDS_FILL_16THS;
            const int64_t need = (g_pool_claims_hint[0] + g_pool_claims_hint[1]) * 11 / 10;      // (+10 %: buckets are not even)
            if (tune.lds_small_buckets == 16 || (tune.lds_small_buckets != 8 && need / 8 > one_pass)) mp->lds_bucket_bits = LG_LDS_BITS_SMALL16;
        }
        const int64_t n_buckets = (int64_t)1 << mp->lds_bucket_bits;
        {
            // one claim list per bucket, twice an even share each (a bucket that outgrows its list is served from the hop's
            // slots instead, kernels_sample.hip)
            mp->claim_cap = (int32_t)(2 * ((slots + n_buckets - 1) / n_buckets) + 256);
            if (tune.lds_claim_cap > 0) mp->claim_cap = tune.lds_claim_cap;      // tests: force the fallback
            const int64_t chunks = ((int64_t)mp->claim_cap + LG_CLAIM_CHUNK - 1) / LG_CLAIM_CHUNK;      // (interleaved by chunk: LanePtrs)
            mp->claim_pairs = (unsigned long long*)d_alloc_space(n_buckets * chunks * LG_CLAIM_CHUNK * sizeof(unsigned long long));
            mp->claim_cnt = (int32_t*)d_alloc_space(n_buckets * LG_CLAIM_CNT_STRIDE * sizeof(int32_t));
            HIP_CALL(hipMemset(mp->claim_cnt, 0, n_buckets * LG_CLAIM_CNT_STRIDE * sizeof(int32_t)));
        }
        if (n_buckets > 64) {        // 256 buckets: {first place, count} per partition tile and bucket (sample_kernel -> place_kernel)
            const int64_t n_parts = (n_super + lg_lds_k_min(mp->lds_bucket_bits) - 1) / lg_lds_k_min(mp->lds_bucket_bits) + 1;
            mp->run_off = (int32_t*)d_alloc_space(n_parts * n_buckets * 2 * sizeof(int32_t));
        }
        // per-bucket lists of the nodes hops 1 .. H-1 add (later hops must recognise them): twice an even share each;
        // a bucket that outgrows its list is served by scanning sampled_ids instead (kernels_sample.hip)
        int64_t listed = 0;
        for (int i = 1; i < hop_num; i++) listed += mp->max_new[i];
        if (listed > 0) {
            mp->known_cap = (int32_t)(2 * ((listed + n_buckets - 1) / n_buckets) + 256);
            if (tune.lds_known_cap > 0) mp->known_cap = tune.lds_known_cap;   // tests: force the scan
            mp->known_pairs = (unsigned long long*)d_alloc_space(n_buckets * mp->known_cap * sizeof(unsigned long long));
            mp->known_cnt = (int32_t*)d_alloc_space(n_buckets * sizeof(int32_t));
            HIP_CALL(hipMemset(mp->known_cnt, 0, n_buckets * sizeof(int32_t)));
        }
    }
    {   // host-visible error word (kernels OR LG_ERR_* into it; reading it costs the host nothing)
        void* h = nullptr;
        void* dptr = nullptr;
        HIP_CALL(hipHostMalloc(&h, 64, hipHostMallocMapped));
        memset(h, 0, 64);
        HIP_CALL(hipHostGetDevicePointer(&dptr, h, 0));
        mp->err_host = (int32_t*)h;
        mp->err_dev = (int32_t*)dptr;
    }
    mp->SetAggSrcId((int32_t*)d_alloc_space(num_ids * sizeof(int32_t)));
    mp->SetAggDstId((int32_t*)d_alloc_space(num_ids * sizeof(int32_t)));
    mp->SetTmpPartIdx((char*)d_alloc_space(num_ids * sizeof(char)));
    mp->SetTmpPartOff((int32_t*)d_alloc_space(num_ids * sizeof(int32_t)));
    const int64_t max_tiles = (mp->max_slots + LG_TILE - 1) / LG_TILE + 1;
    mp->slot_dst = (int32_t*)d_alloc_space((int64_t)mp->max_slots * sizeof(int32_t));
    mp->slot_pos = (int32_t*)d_alloc_space((int64_t)mp->max_slots * sizeof(int32_t));
    if (tune.col_slots != 0) {       // carried feature-cache slots (column slots): per slot of a hop, per node of the batch
        mp->slot_fs = (int32_t*)d_alloc_space((int64_t)mp->max_slots * sizeof(int32_t));
        mp->node_slot = (int32_t*)d_alloc_space(num_ids * sizeof(int32_t));
        lg::fill_value_i32(nullptr, mp->node_slot, LG_FS_UNKNOWN, num_ids);
        lg::fill_value_i32(nullptr, mp->slot_fs, LG_FS_UNKNOWN, mp->max_slots);
        HIP_CALL(hipDeviceSynchronize());
    }
    mp->tile_state = (unsigned long long*)d_alloc_space((max_tiles / LG_SLOTS_PER_LANE + 2) * sizeof(unsigned long long));
    HIP_CALL(hipMemset(mp->tile_state, 0, (size_t)(max_tiles / LG_SLOTS_PER_LANE + 2) * sizeof(unsigned long long)));
    mp->fh_edge = (RowHdr*)d_alloc_space(num_ids * sizeof(RowHdr));
    mp->hop_scratch = (int32_t*)d_alloc_space(HS_WORDS * sizeof(int32_t));
    HIP_CALL(hipMemset(mp->hop_scratch, 0, HS_WORDS * sizeof(int32_t)));
}

// ---- C API ----------------------------------------------------------------------------------
extern "C" LegionGraphStorage* legion_graph_create(int32_t partition_count, int32_t node_num,
                                                   int64_t edge_num, const int64_t* csr_node_index,
                                                   const int32_t* csr_dst_node_ids)
{
    BuildInfo info;
    info.partition_count = partition_count;
    info.total_num_nodes = node_num;
    info.total_edge_num = edge_num;
    info.csr_node_index = const_cast<int64_t*>(csr_node_index);
    info.csr_dst_node_ids = const_cast<int32_t*>(csr_dst_node_ids);
    GraphStorage* g = NewCompleteGraphStorage();
    g->Build(&info);
    return reinterpret_cast<LegionGraphStorage*>(g);
}

// 1 when logical GPU dev samples from the {neighbour id, feature-cache slot} copy of the column array (column slots)
extern "C" int32_t legion_graph_column_slots(const LegionGraphStorage* g_, int32_t dev)
{
    const GraphStorage* g = reinterpret_cast<const GraphStorage*>(g_);
    return (g && dev >= 0 && dev < g->GetPartitionCount() && g->GetCSRXMatrix(dev) != nullptr) ? 1 : 0;
}

// the cached CSR GPU dev holds after a fill (GraphStorage::GraphCache, SS/storage/graph_storage.cu:76-111): device pointers to
// int64 index[capacity + 1] and int32 dst[index[capacity]], or nulls before any fill (introspection for the parity tests)
extern "C" void legion_graph_cached_csr(const LegionGraphStorage* g_, int32_t dev, const int64_t** index_out, const int32_t** dst_out)
{
    const GraphStorage* g = reinterpret_cast<const GraphStorage*>(g_);
    const bool ok = g && dev >= 0 && dev < g->GetPartitionCount();
    if (index_out) *index_out = ok ? g->CachedCSRIndex(dev) : nullptr;
    if (dst_out) *dst_out = ok ? g->CachedCSRDst(dev) : nullptr;
}

extern "C" void legion_graph_destroy(LegionGraphStorage* g_)
{
    GraphStorage* g = reinterpret_cast<GraphStorage*>(g_);
    if (!g) return;
    g->Finalize();
    delete g;
}

extern "C" LegionFeatureStorage* legion_feature_create(int32_t partition_count, int32_t total_num_nodes,
                                                       int32_t float_feature_len,
                                                       const float* all_float_feature)
{
    CompleteFeatureStorage* f = new CompleteFeatureStorage();
    f->Configure(partition_count, total_num_nodes, float_feature_len, const_cast<float*>(all_float_feature));
    return reinterpret_cast<LegionFeatureStorage*>(static_cast<FeatureStorage*>(f));
}

extern "C" void legion_feature_set_ids(LegionFeatureStorage* f_, int32_t dev_id, int32_t mode,
                                       const int32_t* host_ids, const int32_t* host_labels, int32_t count)
{
    FeatureStorage* f = reinterpret_cast<FeatureStorage*>(f_);
    if (!f) { printf("invalid feature storage ptr\n"); return; }
    if (mode < 0 || mode > 2) { printf("invalid mode: %d\n", mode); return; }
    f->SetIds(dev_id, mode, host_ids, host_labels, count);
}

extern "C" void legion_feature_destroy(LegionFeatureStorage* f_)
{
    FeatureStorage* f = reinterpret_cast<FeatureStorage*>(f_);
    if (!f) return;
    f->Finalize();
    delete f;
}

// ---- output arena (legion_core.h PoolArena) ---------------------------------------------------
static thread_local PoolArena* g_pool_arena = nullptr;
void lg_set_pool_arena(PoolArena* arena) { g_pool_arena = arena; }
PoolArena* lg_get_pool_arena() { return g_pool_arena; }
static inline int64_t arena_round(int64_t b) { return (b + 255) & ~(int64_t)255; }
int64_t lg_pool_arena_bytes(int64_t batch_size, int64_t num_ids, int64_t feature_rows, int64_t float_feature_len)
{
    return 3 * arena_round(num_ids * 4) + arena_round(batch_size * 4) + 2 * arena_round(64) +
           arena_round(feature_rows * float_feature_len * 4);
}
static void* arena_take(int64_t bytes)
{
    PoolArena* a = g_pool_arena;
    const int64_t need = arena_round(bytes);
    if (a->used + need > a->bytes) {
        printf("legion_hip: lane arena of %lld bytes is too small (%lld in use, %lld more asked for)\n", (long long)a->bytes,
               (long long)a->used, (long long)need);
        exit(EXIT_FAILURE);
    }
    void* p = a->base + a->used;
    a->used += need;
    return p;
}

extern "C" LegionMemoryPool* legion_pool_create(int32_t dev_id, int32_t total_num_nodes, int32_t batch_size,
                                                const int32_t* fanout, int32_t hop_num,
                                                int32_t float_feature_len, int32_t pipeline_depth)
{
    if (pipeline_depth < 1) pipeline_depth = 1;
    MemoryPool* mp = new MemoryPool(pipeline_depth);
    lg_pool_alloc_private(mp, dev_id, total_num_nodes, batch_size, fanout, hop_num, float_feature_len);
    const bool in_arena = g_pool_arena != nullptr && pipeline_depth == 1;
    mp->owns_buffers = !in_arena;
    mp->arena_backed = in_arena;
    auto take = [&](int64_t bytes) { return in_arena ? arena_take(bytes) : d_alloc_space(bytes); };
    if (in_arena && g_pool_arena->mirror_dev != nullptr && g_pool_arena->mirror_used < g_pool_arena->mirror_lanes) {
        mp->counter_mirror_dev = g_pool_arena->mirror_dev + 32 * (int64_t)g_pool_arena->mirror_used;
        mp->counter_mirror_host = g_pool_arena->mirror_host + 32 * (int64_t)g_pool_arena->mirror_used;
        g_pool_arena->mirror_used++;
    }
    for (int i = 0; i < pipeline_depth; i++) {            // ipc_service.cu:134-161 without the handles
        mp->SetSampledIds((int32_t*)take((int64_t)mp->num_ids * 4), i);
        mp->SetLabels((int32_t*)take((int64_t)batch_size * 4), i);
        mp->SetAggSrcOf((int32_t*)take((int64_t)mp->num_ids * 4), i);
        mp->SetAggDstOf((int32_t*)take((int64_t)mp->num_ids * 4), i);
        mp->SetNodeCounter((int32_t*)take(16 * 4), i);
        mp->SetEdgeCounter((int32_t*)take(16 * 4), i);
        mp->SetCurrentPipe(i);
        HIP_CALL(hipMemset(mp->GetNodeCounter(), 0, 64));
        HIP_CALL(hipMemset(mp->GetEdgeCounter(), 0, 64));
    }
    mp->SetCurrentPipe(0);
    return reinterpret_cast<LegionMemoryPool*>(mp);
}

extern "C" void legion_pool_alloc_features(LegionMemoryPool* p_, int64_t rows)
{
    MemoryPool* mp = reinterpret_cast<MemoryPool*>(p_);
    if (!mp) { printf("invalid memorypool ptr\n"); return; }
    SetGPUDevice(mp->dev_id);
    const int32_t cur = mp->GetCurrentPipe();
    for (int i = 0; i < mp->PipelineDepth(); i++) {
        mp->SetCurrentPipe(i);
        if (mp->arena_backed) {           // (once per pool: the arena is sized for exactly one feature buffer per lane)
            if (g_pool_arena == nullptr) { printf("legion_hip: arena-backed pool without its arena\n"); exit(EXIT_FAILURE); }
            mp->SetFloatFeatures((float*)arena_take(rows * (int64_t)mp->float_feature_len * sizeof(float)), i);
            continue;
        }
        d_free_space(mp->GetFloatFeatures());
        mp->SetFloatFeatures((float*)d_alloc_space(rows * (int64_t)mp->float_feature_len * sizeof(float)), i);
    }
    mp->SetCurrentPipe(cur);
    mp->feature_rows = rows;
}

extern "C" void legion_pool_set_current_pipe(LegionMemoryPool* p_, int32_t pipe)
{
    MemoryPool* mp = reinterpret_cast<MemoryPool*>(p_);
    if (mp) mp->SetCurrentPipe(pipe % mp->PipelineDepth());
}

extern "C" void legion_pool_set_mode_iter(LegionMemoryPool* p_, int32_t mode, int32_t iter)
{
    MemoryPool* mp = reinterpret_cast<MemoryPool*>(p_);
    if (mp) { mp->SetCurrentMode(mode); mp->SetIter(iter); }
}

extern "C" int32_t legion_pool_num_ids(const LegionMemoryPool* p_)
{
    const MemoryPool* mp = reinterpret_cast<const MemoryPool*>(p_);
    return mp ? mp->num_ids : 0;
}

extern "C" void* legion_pool_buffer(LegionMemoryPool* p_, int32_t which)
{
    MemoryPool* mp = reinterpret_cast<MemoryPool*>(p_);
    if (!mp) return nullptr;
    switch (which) {
        case 0: return mp->GetSampledIds();
        case 1: return mp->GetFloatFeatures();
        case 2: return mp->GetLabels();
        case 3: return mp->GetAggSrcOf();
        case 4: return mp->GetAggDstOf();
        case 5: return mp->GetNodeCounter();
        case 6: return mp->GetEdgeCounter();
        case 7: return mp->GetAggSrcId();
        case 8: return mp->GetAggDstId();
        case 9: return mp->GetCacheSearchBuffer();
        case 10: return mp->GetTmpPartIdx();
        case 11: return mp->GetTmpPartOff();
        case 12: return mp->GetPositionMap();      // always null: no per-vertex state in this build
        case 13: return mp->node_slot;             // [num_ids] feature-cache slot carried per node (LG_FS_UNKNOWN = -3: look it up), or null
        default: return nullptr;
    }
}

// LG_ERR_* bits raised by the kernels for this pool since it was created (0 = none); host-visible memory,
// valid once the batch has completed
extern "C" int32_t legion_pool_error(const LegionMemoryPool* p_)
{
    const MemoryPool* mp = reinterpret_cast<const MemoryPool*>(p_);
    return mp ? mp->ErrorBits() : 0;
}

// hash buckets per lane of the first-touch de-duplication: 8, 16, 64 or 256
extern "C" int32_t legion_pool_lds_buckets(const LegionMemoryPool* p_)
{
    const MemoryPool* p = reinterpret_cast<const MemoryPool*>(p_);
    return p ? (1 << p->lds_bucket_bits) : 0;
}

extern "C" int64_t legion_pool_state_bytes(const LegionMemoryPool* p_)
{
    const MemoryPool* mp = reinterpret_cast<const MemoryPool*>(p_);
    if (!mp) return 0;
    // one hop's claim lists + the known lists (nothing that scales with the graph)
    return ((((int64_t)mp->claim_cap + LG_CLAIM_CHUNK - 1) / LG_CLAIM_CHUNK * LG_CLAIM_CHUNK) << mp->lds_bucket_bits) * 8 + ((int64_t)mp->known_cap << mp->lds_bucket_bits) * 8;
}

extern "C" void legion_pool_destroy(LegionMemoryPool* p_)
{
    MemoryPool* mp = reinterpret_cast<MemoryPool*>(p_);
    if (!mp) return;
    mp->Finalize();
    delete mp;
}
