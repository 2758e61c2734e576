"""The dickens-sized workload cut into N independent chunks, one batch (device-resident in/out): usage chunked_dickens.py [N ...]
Prints wall / device ms per stage, MB/s and ratio for every N (chunks are the only way the path shards)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from x3_compressor_amd import _lib, synth
counts = [int(a) for a in sys.argv[1:]] or [64, 128, 256, 512]
data = synth.english_like(synth.DICKENS_BYTES)
n = data.size
dev = torch.device("cuda", 0)
d_in = torch.from_numpy(data).to(dev)
ctx = _lib.X3Context(0)
prm = _lib.make_params(w_kib=64, t=256)
for nch in counts:
    cb = (n + nch - 1) // nch
    off = np.array(list(range(0, n, cb)) + [n], dtype=np.uint64)
    stride = (cb + (cb >> 1) + 4096 + 3) & ~3
    d_out = torch.empty(stride * (len(off) - 1), dtype=torch.uint8, device=dev)
    best = None
    for it in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        lens, st = ctx.compress_chunks_dev(d_in.data_ptr(), off, prm, d_out.data_ptr(), stride)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        if it and (best is None or dt < best[0]): best = (dt, st)
    dt, st = best
    print(f"{len(off)-1:5d} chunks x {cb:7d} B: wall {dt*1e3:7.2f} ms -> {n/dt/1e6:8.1f} MB/s  ratio {n/float(lens.sum()):.4f} | device ms total {st.ms_total:.2f}: copy {st.ms_copy:.2f} scan {st.ms_scan:.2f} parse {st.ms_parse:.2f} "
          f"code {st.ms_code:.2f} (features {st.ms_features:.2f} modes {st.ms_modes:.2f} coder {st.ms_coder:.2f} emit {st.ms_emit:.2f}) pipelined {st.pipelined}", flush=True)
