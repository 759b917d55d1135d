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
// Other revisions lay the table out differently: they are reported as unsupported (return 0) and the caller falls back
// to the transaction count the sampler computes itself (kernels_sample.hip).  profiles/r02/link_counter_probe.txt holds
// the probe this is based on (tools/link_counter_probe.sh).
#include "legion_core.h"

#include <cstring>

extern "C" int32_t legion_link_counters(int32_t dev_id, uint64_t* pcie_bytes, uint64_t* xgmi_bytes)
{
    if (pcie_bytes) *pcie_bytes = 0;
    if (xgmi_bytes) *xgmi_bytes = 0;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count < 1) return 0;
    char bdf[64] = {0};
    const int physical = (legion_get_device_base() + dev_id) % count;
    if (hipDeviceGetPCIBusId(bdf, sizeof(bdf), physical) != hipSuccess) { (void)hipGetLastError(); return 0; }
    for (char* c = bdf; *c; c++) if (*c >= 'A' && *c <= 'F') *c = (char)(*c - 'A' + 'a');
    char path[160];
    snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/gpu_metrics", bdf);
    FILE* f = fopen(path, "rb");
    if (!f) return 0;
    unsigned char buf[4096];
    const size_t n = fread(buf, 1, sizeof(buf), f);
    fclose(f);
    if (n < 264) return 0;
    const unsigned format = buf[2], content = buf[3];
    if (format != 1 || content != 8) return 0;           // only the layout verified on this pool
    uint64_t acc = 0;
    memcpy(&acc, buf + 88, 8);
    if (pcie_bytes) *pcie_bytes = (uint64_t)((double)acc * 102.4);
    uint64_t x = 0;
    for (int l = 0; l < 8; l++) {
        uint64_t r = 0, w = 0;
        memcpy(&r, buf + 136 + 8 * l, 8);
        memcpy(&w, buf + 200 + 8 * l, 8);
        if (r != ~0ull) x += r * 1024;
        if (w != ~0ull) x += w * 1024;
    }
    if (xgmi_bytes) *xgmi_bytes = x;
    return 1;
}
