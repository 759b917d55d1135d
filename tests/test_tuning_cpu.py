"""LegionTuning (include/legion_hip.h section 6): one struct, filled from the LEGION_* environment by one function, or set
by the host program.  Host-only: no GPU needed."""
import ctypes
import os

import pytest

from legion_amd import engine, lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(autouse=True)
def _restore():
    yield
    engine.tuning_from_env()


def test_defaults_and_environment(monkeypatch):
    for k in ("LEGION_LDS_SMALL_BUCKETS", "LEGION_RUNNER_LANES", "LEGION_LINK_COUNTERS", "LEGION_NO_SHM_MIRROR", "LEGION_RUNNER_HANDOVER",
              "LEGION_TABLE_PLACEMENT", "LEGION_RUNNER_HO_STREAM", "LEGION_GATHER_ROWS", "LEGION_RUNNER_SPIN_US"):
        monkeypatch.delenv(k, raising=False)
    engine.tuning_from_env()
    t = engine.tuning()
    assert t["lds_small_buckets"] == 0 and t["lds_part_wg"] == 8192 and t["sample_max_wg"] == 4096 and t["runner_graph"] == 1
    assert t["runner_ho_stream"] == 2 and t["shm_mirror"] == 1 and t["link_counters"] == 0 and t["table_placement"] == 0
    assert t["gather_rows_per_wg"] == 0 and t["col_slots"] == -1 and t["runner_spin_us"] == -1 and t["runner_handover"] == 0
    assert len(t) <= 26                                         # (the struct had grown to 38 switches)
    monkeypatch.setenv("LEGION_LDS_SMALL_BUCKETS", "16")
    monkeypatch.setenv("LEGION_RUNNER_HANDOVER", "gather")
    monkeypatch.setenv("LEGION_RUNNER_LANES", "3")
    monkeypatch.setenv("LEGION_LINK_COUNTERS", "123,45")
    monkeypatch.setenv("LEGION_NO_SHM_MIRROR", "1")
    monkeypatch.setenv("LEGION_TABLE_PLACEMENT", "pinned")
    engine.tuning_from_env()
    t = engine.tuning()
    assert (t["lds_small_buckets"], t["runner_handover"], t["runner_lanes"], t["shm_mirror"], t["table_placement"]) == (16, 1, 3, 0, 1)
    assert t["link_counters"] == 3 and t["link_counter_values"] == [123, 45]
    for word, code in (("v2", 0), ("measured", 1), ("smi", 2)):
        monkeypatch.setenv("LEGION_LINK_COUNTERS", word)
        engine.tuning_from_env()
        assert engine.tuning()["link_counters"] == code


def test_programmatic_values_survive_until_the_environment_is_asked_again(monkeypatch):
    monkeypatch.delenv("LEGION_LDS_SMALL_BUCKETS", raising=False)
    engine.tuning_from_env()
    engine.set_tuning(lds_small_buckets=16, runner_lanes=7)
    t = engine.tuning()
    assert t["lds_small_buckets"] == 16 and t["runner_lanes"] == 7 and t["lds_part_wg"] == 8192     # the rest untouched
    monkeypatch.setenv("LEGION_LDS_SMALL_BUCKETS", "8")
    assert engine.tuning()["lds_small_buckets"] == 16           # installed values are kept ...
    engine.tuning_from_env()
    assert engine.tuning()["lds_small_buckets"] == 8            # ... until the environment is asked for explicitly


def test_struct_mirror_matches_the_header():
    """The ctypes mirror has the header's field order and size."""
    import re
    hdr = open(os.path.join(ROOT, "include", "legion_hip.h")).read()
    body = hdr[hdr.index("typedef struct LegionTuning {"):hdr.index("} LegionTuning;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)                         # (comments may mention types)
    fields = re.findall(r"\b(int32_t|uint64_t)\s+(\w+)(\[\d+\])?\s*;", body)
    assert [n for _, n, _ in fields] == [n for n, _ in lib.Tuning._fields_]     # same names, same order
    n32 = sum(1 for t, _, _ in fields if t == "int32_t")
    assert fields[-1][0] == "uint64_t" and n32 == len(fields) - 1
    assert ctypes.sizeof(lib.Tuning) == (n32 * 4 + 7) // 8 * 8 + 16             # int32 fields, padding to 8, 2 uint64
    assert ctypes.sizeof(lib.LinkCounters) == 8 * 3 + 8 * 16 + 8 + 32 + 8


def test_integration_table_is_generated_from_the_header():
    """INTEGRATION.md's LegionTuning table is tools/tuning_table.py's rendering of include/legion_hip.h."""
    import subprocess
    import sys
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "tuning_table.py"), "--check"], stdout=subprocess.PIPE, text=True)
    assert res.returncode == 0, res.stdout


def test_no_stray_environment_reads_in_the_library():
    """Every LEGION_* variable the library reads is parsed in tuning.hip (LegionTuning), except the deployment names (which shm /
    semaphore namespace, whether this process owns the names): VERDICT r04 counted 12 stray getenv calls."""
    import glob
    import re
    stray = []
    for f in glob.glob(os.path.join(ROOT, "legion_amd", "csrc", "*")):
        if f.endswith("tuning.hip"):
            continue
        for m in re.finditer(r'getenv\("([A-Z_0-9]+)"\)', open(f).read()):
            if m.group(1) not in ("LEGION_IPC_NAMESPACE", "LEGION_IPC_LOCAL", "HSA_ENABLE_IPC_MODE_LEGACY"):
                stray.append((os.path.basename(f), m.group(1)))
    assert not stray, stray


def test_link_counters_without_a_gpu_report_unsupported():
    c = engine.link_counters_ex(0)
    assert set(c) >= {"supported", "pcie_bytes", "xgmi_read_bytes", "xgmi_read_bytes_link", "gpu_metrics_revision", "pci_bus_id"}
    assert len(c["xgmi_read_bytes_link"]) == 8
