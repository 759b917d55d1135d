"""Times legion_gather_rows alone (HIP events on the launch stream) for the variants selected by
LEGION_GATHER_VARIANT: rows random in a table far larger than the Infinity Cache."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from legion_amd import lib, synth
L = lib.load()
N, D = 1 << 25, 128                     # 17 GB table
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 180000
table = synth.features_device(N, D, 7)
g = torch.Generator(device="cuda"); g.manual_seed(1)
ids = torch.randint(0, N, (rows,), device="cuda", dtype=torch.int32, generator=g)
dst = torch.empty((rows, D), device="cuda"); cidx = torch.empty(rows, dtype=torch.int32, device="cuda")
rng = torch.tensor([0, rows], dtype=torch.int32, device="cuda")
p = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def run():
    L.legion_gather_rows(st, p(table), None, None, 1, D, N, p(ids), p(cidx), p(rng), p(dst), rows)
for _ in range(5): run()
torch.cuda.synchronize()
ts = []
for _ in range(30):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); run(); b.record(); b.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
ts.sort()
us = ts[len(ts) // 2]
print(f"variant {os.environ.get('LEGION_GATHER_VARIANT', '0')}: rows={rows} median {us:.1f} us  -> {rows * (8 * D + 8) / us / 1e3:.0f} GB/s algorithmic ({rows * (8 * D + 8) / us / 1e3 / 8000 * 100:.1f}% of 8 TB/s)")
assert bool((dst == table[ids.long()]).all())
