"""Times legion_gather_rows alone (HIP events on the launch stream): rows drawn uniformly from tables of several sizes.
    python3 tools/gather_probe.py <rows> <D> <log2 N> [<log2 N> ...] [--sorted] [--lanes L]
What the feature gather reaches when every source row is cold and uniformly spread (no hot set, no reuse): the
ceiling for the last hop of deep fan-outs, whose new nodes are mostly low-degree vertices all over the table."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from legion_amd import lib, synth
L = lib.load()
args = [a for a in sys.argv[1:] if not a.startswith("--")]
rows, D = int(args[0]), int(args[1])
srt = "--sorted" in sys.argv
p = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for lg2 in [int(a) for a in args[2:]]:
    N = 1 << lg2
    table = synth.features_device(N, D, 7)
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    ids = torch.randint(0, N, (rows,), device="cuda", dtype=torch.int32, generator=g)
    if srt: ids = torch.sort(ids).values
    dst = torch.empty((rows, D), device="cuda"); cidx = torch.empty(rows, dtype=torch.int32, device="cuda")
    rng = torch.tensor([0, rows], dtype=torch.int32, device="cuda")
    def run():
        L.legion_gather_rows(st, p(table), None, None, 1, D, N, p(ids), p(cidx), p(rng), p(dst), rows)
    for _ in range(3): run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(15):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); run(); b.record(); b.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    us = ts[len(ts) // 2]
    gb = rows * (8 * D + 8) / us / 1e3
    print(f"table 2^{lg2} x {D} ({N * D * 4 / 2**30:.1f} GiB) rows={rows}{' sorted' if srt else ''}: median {us:.1f} us -> {gb:.0f} GB/s algorithmic ({gb / 80:.1f}% of 8 TB/s)", flush=True)
    if lg2 <= 22: assert bool((dst == table[ids.long()]).all())
    del table, dst, ids
    torch.cuda.empty_cache()
