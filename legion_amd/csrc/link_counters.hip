// link_counters.hip -- cumulative PCIe / xGMI byte counters of a GPU, read from the driver's gpu_metrics table.
//
// The paper's cost model is fed with the PCIe transactions Intel PCM counted during the PreSC epoch
// (SS/engine/server.cu:105-110, SS/engine/monitor.cuh; hard-wired to {0,0} in v2).  On an MI355X node the counterpart
// is the PMFW metrics table the amdgpu driver exposes, world-readable, at /sys/bus/pci/devices/<bdf>/gpu_metrics
// (rocm-smi / amd-smi decode the same table but print only the instantaneous PCIe bandwidth).  In revision 1.8:
//   offset  88  uint64 pcie_bandwidth_acc      sum over 1-ms cycles of the PCIe bandwidth in 100 KiB/s units
//                                              => bytes = value * 102.4   (measured here: a 32 GiB host->device copy
//                                              moved the counter by 334 955 892 = 34.30e9 bytes, -0.2 %)
//   offset 136  uint64 xgmi_read_data_acc[8]   KiB read over each xGMI link
//   offset 200  uint64 xgmi_write_data_acc[8]  KiB written over each xGMI link
// Other revisions lay the table out differently.  Round 4: the primary source is therefore rocm_smi_lib's own decoder of the
// same table (rsmi_dev_gpu_metrics_info_get, a versioned struct: below), and the byte offsets above are the cross-check and the
// fallback when the library is absent; a table neither can decode is reported as unsupported (return 0, with the revision found) and
// the caller decides: the server refuses to start when the link's counters were asked for (LEGION_LINK_COUNTERS=smi) and
// cannot be had; bench.py falls back to the transaction counts the kernels compute themselves.  profiles/r02/link_counter_probe.txt holds
// the probe this is based on (tools/link_counter_probe.sh).
#include "legion_core.h"

#include <dlfcn.h>
#include <rocm_smi/rocm_smi.h>      // types only: the library is opened at run time (this library must load without it)

#include <cstring>
#include <mutex>

// ---- source 1: rocm_smi_lib's versioned decoder -------------------------------------------------------------------------
// rsmi_dev_gpu_metrics_info_get decodes whatever revision of the table the driver exposes into one struct
// (rsmi_gpu_metrics_t: pcie_bandwidth_acc, xgmi_read_data_acc[8], xgmi_write_data_acc[8]), so a driver update that moves the
// fields does not turn LEGION_LINK_COUNTERS=smi into a refusal.  The library is dlopen'ed (librocm_smi64.so.1).
namespace {
struct Rsmi {
    void* h = nullptr;
    rsmi_status_t (*init)(uint64_t) = nullptr;
    rsmi_status_t (*num)(uint32_t*) = nullptr;
    rsmi_status_t (*pci)(uint32_t, uint64_t*) = nullptr;
    rsmi_status_t (*metrics)(uint32_t, rsmi_gpu_metrics_t*) = nullptr;
    bool ok = false, tried = false;
};
Rsmi g_rsmi;
std::mutex g_rsmi_mu;

bool rsmi_ready()
{
    std::lock_guard<std::mutex> lk(g_rsmi_mu);
    if (g_rsmi.tried) return g_rsmi.ok;
    g_rsmi.tried = true;
    // RTLD_DEEPBIND: the library must resolve its OWN symbols first.  This process also holds libamd_smi (a dependency of RCCL),
    // which carries the same rocm_smi C++ classes under the same names; bound to those, rsmi_* calls corrupted the heap
    // ("corrupted size vs. prev_size" at exit, tools/smi_probe.py)
    const int flags = RTLD_NOW | RTLD_LOCAL | RTLD_DEEPBIND;
    g_rsmi.h = dlopen("librocm_smi64.so.1", flags);
    if (!g_rsmi.h) g_rsmi.h = dlopen("librocm_smi64.so", flags);
    if (!g_rsmi.h) return false;
    g_rsmi.init = (rsmi_status_t(*)(uint64_t))dlsym(g_rsmi.h, "rsmi_init");
    g_rsmi.num = (rsmi_status_t(*)(uint32_t*))dlsym(g_rsmi.h, "rsmi_num_monitor_devices");
    g_rsmi.pci = (rsmi_status_t(*)(uint32_t, uint64_t*))dlsym(g_rsmi.h, "rsmi_dev_pci_id_get");
    g_rsmi.metrics = (rsmi_status_t(*)(uint32_t, rsmi_gpu_metrics_t*))dlsym(g_rsmi.h, "rsmi_dev_gpu_metrics_info_get");
    if (!g_rsmi.init || !g_rsmi.num || !g_rsmi.pci || !g_rsmi.metrics) return false;
    if (g_rsmi.init(0) != RSMI_STATUS_SUCCESS) return false;
    g_rsmi.ok = true;
    return true;
}

// "0000:c5:00.0" -> rocm_smi's bdfid ((domain << 32) | (bus << 8) | (device << 3) | function; the partition bits stay 0)
bool parse_bdf(const char* s, uint64_t* out)
{
    unsigned dom = 0, bus = 0, dev = 0, fn = 0;
    if (sscanf(s, "%x:%x:%x.%x", &dom, &bus, &dev, &fn) != 4) return false;
    *out = ((uint64_t)dom << 32) | ((uint64_t)(bus & 0xff) << 8) | ((uint64_t)(dev & 0x1f) << 3) | (uint64_t)(fn & 0x7);
    return true;
}

void fill_from_raw(LegionLinkCounters* out, uint64_t pcie_acc, const uint64_t* rd, const uint64_t* wr)
{
    out->pcie_bytes = (uint64_t)((double)pcie_acc * 102.4);
    for (int l = 0; l < 8; l++) {
        out->xgmi_read_bytes_link[l] = rd[l] != ~0ull ? rd[l] * 1024 : 0;     // all-ones = link not populated
        out->xgmi_write_bytes_link[l] = wr[l] != ~0ull ? wr[l] * 1024 : 0;
        out->xgmi_read_bytes += out->xgmi_read_bytes_link[l];
        out->xgmi_write_bytes += out->xgmi_write_bytes_link[l];
    }
}

bool read_rsmi(const char* bdf, LegionLinkCounters* out)
{
    if (!rsmi_ready()) return false;
    uint64_t want = 0;
    if (!parse_bdf(bdf, &want)) return false;
    uint32_t n = 0;
    if (g_rsmi.num(&n) != RSMI_STATUS_SUCCESS) return false;
    for (uint32_t i = 0; i < n; i++) {
        uint64_t id = 0;
        if (g_rsmi.pci(i, &id) != RSMI_STATUS_SUCCESS) continue;
        if ((id & 0xFFFFFFFF0000FFFFull) != want) continue;       // (bits 28..31 of the low word: the partition id)
        rsmi_gpu_metrics_t m;
        memset(&m, 0, sizeof(m));
        if (g_rsmi.metrics(i, &m) != RSMI_STATUS_SUCCESS) return false;
        out->format_revision = m.common_header.format_revision;
        out->content_revision = m.common_header.content_revision;
        if (m.pcie_bandwidth_acc == ~0ull) return false;          // this revision of the table does not carry the accumulators
        uint64_t rd[8], wr[8];
        for (int l = 0; l < 8; l++) { rd[l] = m.xgmi_read_data_acc[l]; wr[l] = m.xgmi_write_data_acc[l]; }
        fill_from_raw(out, m.pcie_bandwidth_acc, rd, wr);
        out->source = 1;
        return true;
    }
    return false;
}

// ---- source 2: the table itself, by byte offset (revision 1.8 only: the cross-check, and the fallback without the library) ----
bool read_sysfs(const char* bdf, LegionLinkCounters* out)
{
    char path[160];
    snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/gpu_metrics", bdf);
    FILE* f = fopen(path, "rb");
    if (!f) return false;
    unsigned char buf[4096];
    const size_t n = fread(buf, 1, sizeof(buf), f);
    fclose(f);
    if (n < 4) return false;
    out->format_revision = buf[2];
    out->content_revision = buf[3];
    if (n < 264 || buf[2] != 1 || buf[3] != 8) return false;           // only the layout verified on this pool
    uint64_t acc = 0, rd[8], wr[8];
    memcpy(&acc, buf + 88, 8);
    for (int l = 0; l < 8; l++) {
        memcpy(&rd[l], buf + 136 + 8 * l, 8);
        memcpy(&wr[l], buf + 200 + 8 * l, 8);
    }
    fill_from_raw(out, acc, rd, wr);
    out->source = 2;
    return true;
}
}  // namespace

// Reads the counters of logical GPU dev_id.  source: 0 = rocm_smi_lib's decoder first, the byte-offset parser second; 1 / 2 = that
// source only.  Returns 1 on success; out->format_revision / content_revision say what table was found either way (0/0: none),
// out->source which path produced the numbers.
extern "C" int32_t legion_link_counters_from(int32_t dev_id, int32_t source, LegionLinkCounters* out)
{
    if (!out) return 0;
    memset(out, 0, sizeof(*out));
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count < 1) return 0;
    char bdf[64] = {0};
    const int physical = (legion_get_device_base() + dev_id) % count;
    if (hipDeviceGetPCIBusId(bdf, sizeof(bdf), physical) != hipSuccess) { (void)hipGetLastError(); return 0; }
    for (char* c = bdf; *c; c++) if (*c >= 'A' && *c <= 'F') *c = (char)(*c - 'A' + 'a');
    snprintf(out->pci_bus_id, sizeof(out->pci_bus_id), "%s", bdf);
    if (source != 2 && read_rsmi(bdf, out)) return 1;
    if (source == 1) return 0;
    const int32_t fr = out->format_revision, cr = out->content_revision;
    LegionLinkCounters t;
    memset(&t, 0, sizeof(t));
    snprintf(t.pci_bus_id, sizeof(t.pci_bus_id), "%s", bdf);
    if (read_sysfs(bdf, &t)) { *out = t; return 1; }
    if (t.format_revision != 0) { out->format_revision = t.format_revision; out->content_revision = t.content_revision; }
    else { out->format_revision = fr; out->content_revision = cr; }
    return 0;
}

extern "C" int32_t legion_link_counters_ex(int32_t dev_id, LegionLinkCounters* out) { return legion_link_counters_from(dev_id, 0, out); }

extern "C" int32_t legion_link_counters(int32_t dev_id, uint64_t* pcie_bytes, uint64_t* xgmi_bytes)
{
    if (pcie_bytes) *pcie_bytes = 0;
    if (xgmi_bytes) *xgmi_bytes = 0;
    LegionLinkCounters c;
    if (!legion_link_counters_ex(dev_id, &c)) return 0;
    if (pcie_bytes) *pcie_bytes = c.pcie_bytes;
    if (xgmi_bytes) *xgmi_bytes = c.xgmi_read_bytes + c.xgmi_write_bytes;
    return 1;
}
