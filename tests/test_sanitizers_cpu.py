"""The sanitizer leg of the CPU suite: every piece of plain host C / C++ of the test infrastructure and the
tools, built with AddressSanitizer + UndefinedBehaviorSanitizer (oracle/Makefile target `san`, -fno-sanitize-recover: any finding
ends the program) and run -- the oracle under its own known-answer and property tests, rocThrust's host generator against the
committed golden file, the runner-schedule simulation, and the protocol-only trainer end against a server end played by this test.
CPU only: the GPU box has no sanitizer."""
import ctypes
import json
import mmap
import os
import struct
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = os.path.join(ROOT, "oracle", "_build", "san")
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")


@pytest.fixture(scope="module")
def san():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "san"])
    return SAN


def test_oracle_kats_and_properties_under_asan_ubsan(san):
    """The oracle's own CPU tests in a child interpreter that loads the sanitized library (LEGION_ORACLE_LIB) with the ASan
    runtime preloaded; leak checking off for the interpreter's sake, everything else fatal."""
    libasan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    lib = os.path.join(san, "liblegion_oracle.so")
    env = dict(ENV, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0", LEGION_ORACLE_LIB=lib)
    tests = ["tests/test_oracle_rng.py", "tests/test_oracle_sampler.py", "tests/test_oracle_properties.py", "tests/test_oracle_hybrid.py",
             "tests/test_oracle_dgl_semantics.py", "tests/test_steps.py"]
    check = ("import sys, pytest\n"
             "rc = pytest.main(['-x', '-q', '-p', 'no:cacheprovider'] + sys.argv[1:])\n"
             "maps = open('/proc/self/maps').read()\n"
             f"assert {lib!r} in maps, 'the sanitized oracle was not the library loaded'\n"
             "sys.exit(int(rc))\n")
    res = subprocess.run([sys.executable, "-c", check] + tests, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-4000:]
    assert "passed" in res.stdout and "AddressSanitizer" not in res.stdout and "runtime error" not in res.stdout, res.stdout[-4000:]


def test_thrust_host_generator_under_asan_ubsan_reproduces_the_golden_file(san):
    res = subprocess.run([os.path.join(san, "thrust_pin")], env=ENV, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    assert json.loads(res.stdout) == json.load(open(os.path.join(ROOT, "tests", "golden", "rng_thrust.json")))


def test_runner_schedule_simulation_under_asan_ubsan(san):
    res = subprocess.run([os.path.join(san, "runner_schedule_test")], env=ENV, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert res.returncode == 0 and "VIOLATION" not in res.stdout, res.stdout[-3000:]


# ---- the wire protocol's server end, played in Python for tools/boundary_consumer.c ------------------------------------------------
MAX_DEVICE, INTERBATCH_CON, MEMORY_USAGE = 8, 2, 7
SHM_BYTES = 12 + MAX_DEVICE * INTERBATCH_CON * MEMORY_USAGE * 64                       # simpleIPCshm (SS/engine/ipc_service.cu:28-31)
EXT_COUNTERS = 16                                                                      # legionIPCext: 4 ints, then counters[8][2][32]
EXT_BYTES = 16 + MAX_DEVICE * INTERBATCH_CON * 32 * 4 + MAX_DEVICE * 64 + MAX_DEVICE * 8 + MAX_DEVICE * 4 + \
    MAX_DEVICE * INTERBATCH_CON * 4 + MAX_DEVICE * INTERBATCH_CON * 5 * 8


class Sem:
    _rt = ctypes.CDLL("libpthread.so.0", use_errno=True)
    _rt.sem_open.restype = ctypes.c_void_p
    _rt.sem_open.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_uint, ctypes.c_uint]
    for _f in ("sem_post", "sem_close"):
        getattr(_rt, _f).argtypes = [ctypes.c_void_p]
    _rt.sem_timedwait.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    _rt.sem_unlink.argtypes = [ctypes.c_char_p]

    def __init__(self, name):
        self.name = name.encode()
        self.h = self._rt.sem_open(self.name, os.O_CREAT | os.O_RDWR, 0o666, 0)
        assert self.h not in (None, 0), f"sem_open {name}"

    def post(self):
        assert self._rt.sem_post(self.h) == 0

    def wait(self, seconds=30):
        import time
        t = time.clock_gettime(time.CLOCK_REALTIME) + seconds
        ts = (ctypes.c_long * 2)(int(t), int((t % 1) * 1e9))
        assert self._rt.sem_timedwait(self.h, ts) == 0, f"{self.name}: the trainer end never released its slot"

    def close(self):
        self._rt.sem_close(self.h)
        self._rt.sem_unlink(self.name)


@pytest.mark.parametrize("views", [0, 1])
def test_protocol_only_trainer_end_under_asan_ubsan(san, views):
    """Slab + mirror object + four named semaphores as ipc_env.hip creates them; batches posted in pipe order 0, 1, 0, ...
    (SS/engine/ipc_service.cu:181-192,283-291).  The consumer must count exactly the training batches' edges."""
    ns = f"_san{os.getpid()}_{views}"
    dev, hops, steps, epochs = 1, 2, (5, 2, 3), 2
    names = [f"/dev/shm/simpleIPCshm{ns}", f"/dev/shm/legionIPCext{ns}"]
    sems = []
    try:
        with open(names[0], "wb") as f:
            f.write(struct.pack("<3i", *steps) + bytes(SHM_BYTES - 12))
        with open(names[1], "wb") as f:
            f.write(struct.pack("<4i", 0x4C47494F, 3, 0, 0) + bytes(EXT_BYTES - 16))
        ext_f = open(names[1], "r+b")
        ext = mmap.mmap(ext_f.fileno(), EXT_BYTES)
        arena_bytes_off = 16 + MAX_DEVICE * INTERBATCH_CON * 32 * 4 + MAX_DEVICE * 64
        struct.pack_into("<q", ext, arena_bytes_off + 8 * dev, 1 << 20)          # an arena is published for this GPU
        sr = [Sem(f"sem_r_{dev}_{i}{ns}") for i in range(2)]
        sw = [Sem(f"sem_w_{dev}_{i}{ns}") for i in range(2)]
        sems = sr + sw
        proc = subprocess.Popen([os.path.join(san, "boundary_consumer"), ns, str(dev), str(hops), "0", str(epochs), str(views)],
                                env=ENV, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        per_epoch = steps[0] + steps[1]
        total, train = per_epoch * epochs + steps[2], per_epoch * (epochs - 1) + steps[0]
        want_edges = want_nodes = 0
        pipe = 0
        for i in range(total):
            sr[pipe].wait()
            edges, nodes = 1000 + 7 * i, 300 + i
            base = EXT_COUNTERS + ((dev * INTERBATCH_CON + pipe) * 32) * 4
            struct.pack_into("<i", ext, base + (9 + hops) * 4, nodes)                 # node_counter[9 + H]
            struct.pack_into("<i", ext, base + (16 + 9 + hops) * 4, edges)            # edge_counter[9 + H]
            if i < train:
                want_edges += edges
                want_nodes += nodes
            sw[pipe].post()
            pipe ^= 1
        out, err = proc.communicate(timeout=60)
        assert proc.returncode == 0, err[-3000:]
        line = json.loads(out)
        assert line["views"] == views and line["timed_batches"] == train
        assert abs(line["nodes_per_batch"] - want_nodes / train) < 0.06
        assert abs(line["edges_per_sec"] / line["batches_per_sec"] - want_edges / train) < 1.0
        direct_off = arena_bytes_off + MAX_DEVICE * 8 + 4 * dev
        assert struct.unpack_from("<i", ext, direct_off)[0] == views              # it announced what kind of trainer end it is
    finally:
        for s in sems:
            s.close()
        for n in names:
            if os.path.exists(n):
                os.unlink(n)
