"""bench.py's one-line guard (tools/bench_legs.py: OneLine + tools/exit_line.c), host only: once armed, the headline's line goes out
whatever ends the process -- a native exit() (the C atexit handler holding the serialised line), SIGTERM, a hang -- exactly once; and a
process that prints its line itself leaves through the interpreter's NORMAL exit (a profiler's own exit handlers run: rocprofv3 lost
its counter files to the os._exit() the earlier, Python-callable hook forced)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROLOGUE = ("import os, sys, ctypes, signal, time, atexit\n"
            f"sys.path.insert(0, {ROOT!r})\n"
            "from tools import bench_legs as legs\n"
            "g = legs.OneLine(os.dup(1))\n"
            "line = {'metric': 'm', 'value': 1.5, 'roofline': {'frac': 0.5}}\n"
            "g.arm(line, 3)\n"
            "assert g._native is not None and not g.needs_hard_exit\n"
            "line['other_shapes'] = [{'value': 2}]\n"
            "g.refresh()\n")


def run(body):
    return subprocess.run([sys.executable, "-c", PROLOGUE + body], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)


@pytest.mark.parametrize("how,body,rc", [
    ("native exit", "ctypes.CDLL(None).exit(3)\n", 3),
    ("sigterm", "os.kill(os.getpid(), signal.SIGTERM)\ntime.sleep(30)\n", 1),
    ("hang", "ctypes.CDLL(None).sleep(30)\n", 1),
])
def test_an_armed_guard_prints_the_line_whatever_ends_the_process(how, body, rc):
    res = run(body)
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and res.returncode == rc, (how, res.returncode, res.stdout, res.stderr[-1500:])
    d = json.loads(lines[0])
    assert d["value"] == 1.5 and d["other_shapes"] == [{"value": 2}] and "extra_legs_error" in d       # the refreshed object, with the note


def test_a_process_that_prints_its_line_leaves_normally():
    res = run("atexit.register(lambda: sys.stderr.write('python atexit ran\\n'))\n"
              "line['cpu_baseline'] = {'value': 3}\n"
              "g.emit(line)\n"
              "g.emit(line)\n")                      # (a second emit is a no-op)
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and res.returncode == 0, (res.returncode, res.stdout, res.stderr[-1500:])
    d = json.loads(lines[0])
    assert "extra_legs_error" not in d and d["cpu_baseline"] == {"value": 3}
    assert "python atexit ran" in res.stderr          # the interpreter finalised normally; the C handler had nothing left to print
