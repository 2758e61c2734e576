"""experiment: does the order in which the handle and torch's first device work are created decide how well the masked streams separate? (bench.py creates the handle first)"""
import sys, time, os
sys.path.insert(0, '.')
import numpy as np, torch
from x3_compressor_amd import _lib, synth
order = sys.argv[1] if len(sys.argv) > 1 else "ctx-first"
extra = sys.argv[2] if len(sys.argv) > 2 else ""
data = synth.english_like(synth.DICKENS_BYTES)
n = data.size
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
if extra == "dist": from x3_compressor_amd import dist as xdist
if order == "ctx-first":
    ctx = _lib.X3Context(0); d_in = torch.from_numpy(data).to(dev)
else:
    d_in = torch.from_numpy(data).to(dev); ctx = _lib.X3Context(0)
prm = _lib.make_params(w_kib=64, t=256)
if extra == "steps":
    stride1 = (2 * n + 4096 + 3) & ~3
    d_out1 = torch.empty(stride1, dtype=torch.uint8, device=dev); off1 = np.array([0, n], dtype=np.uint64)
    for _ in range(23): ctx.compress_chunks_dev(d_in.data_ptr(), off1, prm, d_out1.data_ptr(), stride1)
    torch.cuda.synchronize(); ctx.compress(data, prm); ctx.compress(data, prm)
out = []
for nch in (16, 24, 32, 40, 48, 64, 80, 96):
    cb = (n + nch - 1) // nch
    off = np.array(list(range(0, n, cb)) + [n], dtype=np.uint64)
    stride = (cb + (cb >> 1) + 4096 + 3) & ~3
    d_out = torch.empty(stride * (len(off) - 1), dtype=torch.uint8, device=dev)
    best = 1e9
    for it in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ctx.compress_chunks_dev(d_in.data_ptr(), off, prm, d_out.data_ptr(), stride)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        if it: best = min(best, dt)
    out.append(f"{nch}: {best*1e3:.2f}")
print(f"{order:12s} {extra:6s}", "  ".join(out), flush=True)
