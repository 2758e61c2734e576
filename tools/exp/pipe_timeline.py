"""time line of the pipelined schedule (X3H_DEBUG prints where the coder segments sit): config 4 share, config 2"""
import os, sys
os.environ["X3H_DEBUG"] = "1"
sys.path.insert(0, '.')
import numpy as np
from x3_compressor_amd import _lib, synth
ctx = _lib.X3Context(0)
nch, cb = 16, 8 << 20
data = synth.zipf_bytes(nch * cb); off = np.arange(0, (nch + 1) * cb, cb, dtype=np.uint64)
for _ in range(2):
    streams = ctx.compress_chunks(data, off, _lib.make_params(w_kib=64, t=256), stride=cb + (cb >> 2)); st = ctx.last_stats
    print(f"config 4 share: total {st.ms_total:.1f} ms scan {st.ms_scan:.1f} features {st.ms_features:.1f} modes {st.ms_modes:.1f} coder {st.ms_coder:.1f}", flush=True)
d = synth.english_like(synth.DICKENS_BYTES).tobytes()
for _ in range(2):
    s = ctx.compress(d, _lib.make_params(w_kib=64, t=256)); st = ctx.last_stats
    print(f"config 2: total {st.ms_total:.1f} ms scan {st.ms_scan:.1f} features {st.ms_features:.1f} modes {st.ms_modes:.1f} coder {st.ms_coder:.1f}", flush=True)
