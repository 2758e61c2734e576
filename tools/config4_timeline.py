"""config 4's per-GPU share (16 x 8 MiB of the Zipf stream, -w 64 -t 256) under X3H_DEBUG=1: where the slices' stages and coder segments sit on the time line of the call"""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from x3_compressor_amd import _lib, synth
CH = 8 << 20
data = synth.zipf_bytes(16 * CH)
dev = torch.device("cuda", 0)
d_in = torch.from_numpy(data).to(dev)
stride = (CH + (CH >> 2) + 4096 + 3) & ~3
d_out = torch.empty(stride * 16, dtype=torch.uint8, device=dev)
off = np.arange(0, 17 * CH, CH, dtype=np.uint64)
ctx = _lib.X3Context(0)
prm = _lib.make_params(w_kib=64, t=256)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    lens, st = ctx.compress_chunks_dev(d_in.data_ptr(), off, prm, d_out.data_ptr(), stride)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"run {rep}: wall {dt*1e3:.1f} ms, device total {st.ms_total:.1f}: scan {st.ms_scan:.1f} parse {st.ms_parse:.1f} features {st.ms_features:.1f} modes {st.ms_modes:.1f} coder {st.ms_coder:.1f}, slices {st.coder_launches}", flush=True)
