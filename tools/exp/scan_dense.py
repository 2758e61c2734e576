"""experiment: K1 on data with dense classes (usage: scan_dense.py kind total_MiB chunk_KiB)"""
import sys, os, time
sys.path.insert(0, '.')
import numpy as np, torch
from x3_compressor_amd import _lib, synth
kind, total, cb = sys.argv[1], int(sys.argv[2]) << 20, int(sys.argv[3]) << 10
data = {"mr": synth.mr_like, "zipf": synth.zipf_bytes, "zeros": lambda n: np.zeros(n, np.uint8), "text": synth.english_like}[kind](total)
dev = torch.device("cuda", 0)
d_in = torch.from_numpy(data).to(dev)
off = np.arange(0, total + 1, cb, dtype=np.uint64)
stride = (cb + (cb >> 1) + 4096 + 3) & ~3
d_out = torch.empty(stride * (len(off) - 1), dtype=torch.uint8, device=dev)
ctx = _lib.X3Context(0)
prm = _lib.make_params(w_kib=int(sys.argv[4]) if len(sys.argv) > 4 else 64, t=int(sys.argv[5]) if len(sys.argv) > 5 else 256)
for it in range(3):
    if it == 2: os.environ["X3H_DEBUG"] = "1"
    torch.cuda.synchronize(); t0 = time.perf_counter()
    lens, st = ctx.compress_chunks_dev(d_in.data_ptr(), off, prm, d_out.data_ptr(), stride)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{kind} {total>>20} MiB in {len(off)-1} chunks: wall {dt*1e3:.1f} ms -> {total/dt/1e6:.1f} MB/s | scan {st.ms_scan:.1f} parse {st.ms_parse:.1f} code {st.ms_code:.1f} ratio {total/float(lens.sum()):.3f}", flush=True)
