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
// Other revisions lay the table out differently: they are reported as unsupported (return 0, with the revision found) and
// the caller decides: the server refuses to start when the link's counters were asked for (LEGION_LINK_COUNTERS=smi) and
// cannot be had; bench.py falls back to the transaction counts the kernels compute themselves.  profiles/r02/link_counter_probe.txt holds
// the probe this is based on (tools/link_counter_probe.sh).
#include "legion_core.h"

#include <cstring>

// Reads the table of logical GPU dev_id.  Returns 1 when the revision is one whose layout is known (1.8), 0 otherwise;
// out->format_revision / content_revision say what was found either way (0/0: no table at all).
extern "C" int32_t legion_link_counters_ex(int32_t dev_id, LegionLinkCounters* out)
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
    char path[160];
    snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/gpu_metrics", bdf);
    FILE* f = fopen(path, "rb");
    if (!f) return 0;
    unsigned char buf[4096];
    const size_t n = fread(buf, 1, sizeof(buf), f);
    fclose(f);
    if (n < 4) return 0;
    out->format_revision = buf[2];
    out->content_revision = buf[3];
    if (n < 264 || buf[2] != 1 || buf[3] != 8) return 0;           // only the layout verified on this pool
    uint64_t acc = 0;
    memcpy(&acc, buf + 88, 8);
    out->pcie_bytes = (uint64_t)((double)acc * 102.4);
    for (int l = 0; l < 8; l++) {
        uint64_t r = 0, w = 0;
        memcpy(&r, buf + 136 + 8 * l, 8);
        memcpy(&w, buf + 200 + 8 * l, 8);
        out->xgmi_read_bytes_link[l] = r != ~0ull ? r * 1024 : 0;     // all-ones = link not populated
        out->xgmi_write_bytes_link[l] = w != ~0ull ? w * 1024 : 0;
        out->xgmi_read_bytes += out->xgmi_read_bytes_link[l];
        out->xgmi_write_bytes += out->xgmi_write_bytes_link[l];
    }
    return 1;
}

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
