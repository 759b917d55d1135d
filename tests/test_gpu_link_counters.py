"""The PCM replacement: cumulative PCIe / xGMI counters of the GPU.  Primary source: rocm_smi_lib's versioned decoder
of the driver's gpu_metrics table (rsmi_dev_gpu_metrics_info_get, opened at run time); cross-check and fallback: the table parsed
by byte offset (revision 1.8).  Reference role: SS/engine/monitor.cuh, SS/engine/server.cu:105-110 (Intel PCM, dead in v2)."""
import time

import pytest
import torch

from legion_amd import engine

pytestmark = pytest.mark.gpu


def test_both_decoders_agree_on_a_4_gib_copy(hip):
    lib0, raw0 = engine.link_counters_ex(0, source=1), engine.link_counters_ex(0, source=2)
    if not lib0["supported"]:
        pytest.skip(f"rocm_smi_lib cannot decode this GPU's gpu_metrics table (revision {lib0['gpu_metrics_revision']})")
    assert lib0["source"] == "rocm_smi_lib" and lib0["pci_bus_id"]
    auto = engine.link_counters_ex(0)
    assert auto["supported"] and auto["source"] == "rocm_smi_lib"       # the versioned decoder is the primary source
    host = torch.empty(1 << 30, dtype=torch.uint8).pin_memory()          # 1 GiB pinned, copied four times
    dev = torch.empty(1 << 30, dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    time.sleep(0.05)
    lib0, raw0 = engine.link_counters_ex(0, source=1), engine.link_counters_ex(0, source=2)
    for _ in range(4):
        dev.copy_(host, non_blocking=True)
    torch.cuda.synchronize()
    time.sleep(0.05)                                                     # the table is refreshed every millisecond or so
    lib1, raw1 = engine.link_counters_ex(0, source=1), engine.link_counters_ex(0, source=2)
    moved = 4 * (1 << 30)
    d_lib = lib1["pcie_bytes"] - lib0["pcie_bytes"]
    assert 0.9 * moved < d_lib < 1.25 * moved, (d_lib, moved)            # the link carried the copy (+ protocol overhead, other traffic)
    if raw0["supported"]:                                                # revision 1.8: the two decoders read the same accumulator
        d_raw = raw1["pcie_bytes"] - raw0["pcie_bytes"]
        assert abs(d_raw - d_lib) <= 0.02 * moved, (d_raw, d_lib)
        assert raw1["gpu_metrics_revision"] == lib1["gpu_metrics_revision"]
        assert raw1["xgmi_read_bytes_link"] == lib1["xgmi_read_bytes_link"]
