"""experiment: what in bench.py's process makes the mid-size batches slower than in tools/chunked_dickens.py?  The same sweep after each of bench.py's earlier steps."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from x3_compressor_amd import _lib, synth
data = synth.english_like(synth.DICKENS_BYTES)
n = data.size
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
d_in = torch.from_numpy(data).to(dev)
ctx = _lib.X3Context(0)
prm = _lib.make_params(w_kib=64, t=256)
def sweep(tag):
    out = []
    for nch in (64, 96):
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
        out.append(f"{nch}: {best*1e3:.2f} ms")
    print(f"{tag:50s}", "  ".join(out), flush=True)
sweep("fresh process")
stride1 = (2 * n + 4096 + 3) & ~3
d_out1 = torch.empty(stride1, dtype=torch.uint8, device=dev)
off1 = np.array([0, n], dtype=np.uint64)
for _ in range(3): ctx.compress_chunks_dev(d_in.data_ptr(), off1, prm, d_out1.data_ptr(), stride1)
torch.cuda.synchronize()
sweep("after three single-stream steps (device buffers)")
ctx.compress(data, prm); ctx.compress(data, prm)
sweep("after two host-buffer compress calls")
s = ctx.compress(data[:1 << 20], prm); ctx.decompress(s, (1 << 20) + 8)
sweep("after a decode")
x = torch.randn(4096, 4096, device=dev); y = x @ x; torch.cuda.synchronize()
sweep("after a torch matmul")
