#!/usr/bin/env python3
"""Where a dedup_lds_kernel workgroup's time goes: runs bench.py with a library built with -DLG_DEDUP_STAMPS
(tools/lds_tuning/build_variant.sh stamps -DLG_DEDUP_STAMPS) and prints, per hop, the mean time from a workgroup's start to
each stamp (thread 0's clock, 100 MHz).   LEGION_HIP_LIB=tools/lds_tuning/variants/stamps/liblegion_hip.so python tools/lds_tuning/dedup_stamps.py [bench args]"""
import ctypes, os, runpy, sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
NAMES = ["", "segment offsets loaded (barrier)", "segment prefix (barrier)", "claim loads issued", "known vertices inserted",
         "claims inserted", "barrier", "look-ups + loser stores", "barrier (end)"]


def dump():
    from legion_amd import lib as L
    lib = ctypes.CDLL(L.LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    buf = (ctypes.c_ulonglong * (8 * 16))()
    lib.legion_debug_dedup_stamps(buf, 0)
    for hop in range(8):
        r = buf[hop * 16:(hop + 1) * 16]
        n = r[0]
        if n == 0:
            continue
        print(f"hop {hop}: {n} workgroups, {r[9] / n:.0f} claims and {r[10] / n:.2f} passes per workgroup", file=sys.stderr)
        prev = 0.0
        for i in range(1, 9):
            t = r[i] / n / 100.0      # 100 MHz -> us
            print(f"   {t:8.2f} us (+{t - prev:6.2f})  {NAMES[i]}", file=sys.stderr)
            prev = t


real_exit = os._exit
def _exit(rc):
    try:
        dump()
    finally:
        sys.stderr.flush()
        real_exit(rc)
os._exit = _exit
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[1:]
runpy.run_path(sys.argv[0], run_name="__main__")
dump()
