"""GPU box: builds synth.csr_device_large(N, E) and checks it against a recomputation of the same chunks:
column range, indptr, and an order-independent checksum of the (row, neighbour) pairs.   python tools/large_graph_check.py N E"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from legion_amd import synth

N, E = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda:0")
t0 = time.time()
indptr, col = synth.csr_device_large(N, E, 20231, dev)
torch.cuda.synchronize()
print(f"built N={N} E={E} in {time.time() - t0:.1f} s; indptr[-1]={int(indptr[-1])}, col range [{int(col.min())}, {int(col.max())}]", flush=True)
assert int(indptr[-1]) == E and int(col.min()) >= 0 and int(col.max()) < N and bool((indptr[1:] >= indptr[:-1]).all())
K1, K2 = 0x9E3779B97F4A7C15 - (1 << 64), 0x2545F4914F6CDD1D          # int64 wrap-around arithmetic
def pair_sum(r, c):
    r, c = r.long(), c.long()
    return int((r * K1 + c * K2 + r * c).sum())
want = 0
import ctypes
from legion_amd import lib as _libmod
L = _libmod.load()
scale = max(1, (N - 1).bit_length())
chunk = 1 << 27
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for ci in range((E + chunk - 1) // chunk):
    n = min(chunk, E - ci * chunk)
    src = torch.empty(n, dtype=torch.int32, device=dev); dst = torch.empty(n, dtype=torch.int32, device=dev)
    cseed = (20231 + 0x9E3779B97F4A7C15 * (ci + 1)) & 0xFFFFFFFFFFFFFFFF
    L.legion_synth_rmat_edges_scrambled(stream, scale, n, cseed, ctypes.c_void_p(src.data_ptr()), ctypes.c_void_p(dst.data_ptr()), synth.SCRAMBLE_KEY)
    if N != (1 << scale):
        src.remainder_(N); dst.remainder_(N)
        loop = src == dst
        dst[loop] = (dst[loop] + 1) % N
    want = (want + pair_sum(src, dst)) & 0xFFFFFFFFFFFFFFFF
got = 0
rows_per = 1 << 22
for r0 in range(0, N, rows_per):
    r1 = min(N, r0 + rows_per)
    deg = indptr[r0 + 1:r1 + 1] - indptr[r0:r1]
    rows = torch.repeat_interleave(torch.arange(r0, r1, device=dev), deg)
    got = (got + pair_sum(rows, col[int(indptr[r0]):int(indptr[r1])])) & 0xFFFFFFFFFFFFFFFF
print("pair checksum", hex(want), hex(got), "OK" if want == got else "MISMATCH", flush=True)
assert want == got
