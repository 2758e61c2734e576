"""experiment: time of x3_ctxseg_kernel with parts of the tile body disabled (wrong results; timing only): usage ctx_variants.py lib.so"""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from x3_compressor_amd import _lib, synth
total, cb = 256 << 20, 256 << 10
data = np.tile(synth.english_like(8 << 20), total // (8 << 20))
off = np.arange(0, total + 1, cb, dtype=np.uint64)
dev = torch.device("cuda", 0)
d_in = torch.from_numpy(data).to(dev)
stride = (cb + (cb >> 1) + 4096 + 3) & ~3
d_out = torch.empty(stride * (len(off) - 1), dtype=torch.uint8, device=dev)
ctx = _lib.X3Context(0, library=sys.argv[1] if len(sys.argv) > 1 else None)
prm = _lib.make_params(w_kib=64, t=256)
for it in range(3):
    try:
        lens, st = ctx.compress_chunks_dev(d_in.data_ptr(), off, prm, d_out.data_ptr(), stride)
        print(f"{sys.argv[1] if len(sys.argv) > 1 else 'product'}: features {st.ms_features:.1f} ms total {st.ms_total:.1f}", flush=True)
    except Exception as e:
        print("error", e); break
