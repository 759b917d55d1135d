"""The Runner's lane-reuse rule (legion_amd/csrc/runner_schedule.h: which launch group goes into which pipeline slot, and when a
slot's lanes may be overwritten) is host-only logic: tests/cpu/runner_schedule_test.cpp drives it against a simulated trainer that
releases as late as the two-slot semaphore protocol allows -- 2 / 3 / 4 groups in flight, 1 ... 64 lanes, full and ragged groups --
and checks that a lane is never overwritten while the trainer may still read the batch that lives in it (the `views` hand-over
reads batches in place).  Compiled with g++, no GPU, no HIP."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_lane_reuse_rule_against_a_simulated_trainer(tmp_path):
    exe = str(tmp_path / "runner_schedule_test")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", os.path.join(ROOT, "tests", "cpu", "runner_schedule_test.cpp"), "-o", exe])
    res = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert res.returncode == 0 and "0 failed" in res.stdout, res.stdout[-2000:]


def test_the_rule_is_tight(tmp_path):
    """One batch more lenient (a slot free once its group ended at or before batch k instead of k-1) and the simulation catches a
    lane overwritten under the trainer: the test above is able to fail."""
    hdr = open(os.path.join(ROOT, "legion_amd", "csrc", "runner_schedule.h")).read()
    assert "return e < 0 || e <= k - 1;" in hdr
    (tmp_path / "runner_schedule.h").write_text(hdr.replace("return e < 0 || e <= k - 1;", "return e < 0 || e <= k;"))
    src = open(os.path.join(ROOT, "tests", "cpu", "runner_schedule_test.cpp")).read().replace("../../legion_amd/csrc/runner_schedule.h", "runner_schedule.h")
    (tmp_path / "t.cpp").write_text(src)
    exe = str(tmp_path / "t")
    subprocess.check_call(["g++", "-O1", "-std=c++17", str(tmp_path / "t.cpp"), "-o", exe], cwd=tmp_path)
    res = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert res.returncode != 0 and "VIOLATION" in res.stdout
