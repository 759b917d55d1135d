#!/bin/bash
# What cumulative PCIe / xGMI counters can an unprivileged user read on this box?  (the reference feeds its
# cost model with Intel PCM PCIe transaction counts, SS/engine/server.cu:105-110, SS/engine/monitor.cuh.)
# Dumps the vendor tools' views and the raw gpu_metrics table before and after a burst of PCIe reads from the GPU.
OUT=${1:-gpurun_out/link_probe}
mkdir -p $OUT
snap() {
  amd-smi metric --pcie --json > $OUT/amdsmi_pcie_$1.json 2>&1
  amd-smi metric --xgmi --json > $OUT/amdsmi_xgmi_metric_$1.json 2>&1
  amd-smi xgmi -m --json > $OUT/amdsmi_xgmi_m_$1.json 2>&1
  rocm-smi --showxgmierr --showpcieinfo --showpciereplaycount --json > $OUT/rocmsmi_$1.json 2>&1
  for c in /sys/class/drm/card*/device; do
    [ -f $c/gpu_metrics ] && od -A d -t u1 -v $c/gpu_metrics > $OUT/gpu_metrics_$(basename $(dirname $c))_$1.txt 2>&1
    [ -f $c/pcie_bw ] && cat $c/pcie_bw > $OUT/pcie_bw_$(basename $(dirname $c))_$1.txt 2>&1
  done
}
snap before
python3 - <<'PY'
import torch, time
# 8 GiB of host-pinned memory read by the GPU over PCIe (the miss path of a pinned-placement run does exactly this)
h = torch.empty(1 << 30, dtype=torch.int64).pin_memory()
d = torch.empty(1 << 30, dtype=torch.int64, device="cuda")
t = time.time()
for _ in range(4):
    d.copy_(h, non_blocking=True)
torch.cuda.synchronize()
print("moved 32 GiB host->device in %.2f s" % (time.time() - t))
PY
snap after
amd-smi metric --help > $OUT/amdsmi_metric_help.txt 2>&1
amd-smi version > $OUT/amdsmi_version.txt 2>&1
ls -la /sys/class/drm/card*/device/ 2>/dev/null | grep -i -E "metrics|pcie|xgmi" > $OUT/sysfs_listing.txt
python3 - $OUT <<'PY'
import glob, sys, os
d = sys.argv[1]
for f in sorted(glob.glob(d + "/gpu_metrics_*_before.txt")):
    g = f.replace("_before", "_after")
    def load(p):
        b = []
        for line in open(p):
            parts = line.split()
            b += [int(x) for x in parts[1:]]
        return bytes(b)
    a, b = load(f), load(g)
    print(os.path.basename(f), "size", len(a), "format", a[2] if len(a) > 3 else None, "content rev", a[3] if len(a) > 3 else None)
    import struct
    diffs = []
    for off in range(0, min(len(a), len(b)) - 7, 8):
        x, y = struct.unpack_from("<Q", a, off)[0], struct.unpack_from("<Q", b, off)[0]
        if x != y and 0 < y - x < (1 << 50): diffs.append((off, x, y, y - x))
    print("  8-byte words that grew:", diffs[:40])
PY
