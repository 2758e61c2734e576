"""config 3 (twelve streams with the Silesia sizes, -w 256 -t 1024) as one batch: wall time under the current X3H_SLICE_MARKS (X3H_DEBUG=1 prints the slices' time line)"""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from x3_compressor_amd import _lib, synth
parts = [synth.config3_part(i) for i in range(len(synth.SILESIA))]
sizes = [int(p.size) for p in parts]
dev = torch.device("cuda", 0)
d_in = torch.from_numpy(np.concatenate(parts)).to(dev)
off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
stride = (max(sizes) + (max(sizes) >> 2) + 4096 + 3) & ~3
d_out = torch.empty(stride * len(sizes), dtype=torch.uint8, device=dev)
prm = _lib.make_params(w_kib=256, t=1024)
ctx = _lib.X3Context(0)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    lens, st = ctx.compress_chunks_dev(d_in.data_ptr(), off, prm, d_out.data_ptr(), stride)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"run {rep}: wall {dt*1e3:.1f} ms: scan {st.ms_scan:.1f} parse {st.ms_parse:.1f} features {st.ms_features:.1f} modes {st.ms_modes:.1f} coder {st.ms_coder:.1f}, slices {st.coder_launches}, bytes {int(lens.sum())}", flush=True)
