"""config 4's share (16 x 8 MiB Zipf) under other checkpoint sets of the pipelined schedule (X3H_PIPE_MARKS is read when the handle is made)"""
import os, sys
sys.path.insert(0, '.')
import numpy as np
from x3_compressor_amd import _lib, synth
nch, cb = 16, 8 << 20
data = synth.zipf_bytes(nch * cb); off = np.arange(0, (nch + 1) * cb, cb, dtype=np.uint64)
prm = _lib.make_params(w_kib=64, t=256)
for marks in (None, "0.18,0.52,0.9", "0.1,0.35,0.7", "0.05,0.2,0.5,0.85", "0.03,0.12,0.35,0.7", "0.02,0.08,0.26,0.62,0.9", "0.12,0.4,0.8"):
    if marks: os.environ["X3H_PIPE_MARKS"] = marks
    with _lib.X3Context(0) as ctx:
        best = 1e9
        for _ in range(3):
            ctx.compress_chunks(data, off, prm, stride=cb + (cb >> 2)); best = min(best, ctx.last_stats.ms_total)
        st = ctx.last_stats
    print(f"marks {marks or 'default 0.02,0.08,0.26,0.62'}: {best:.1f} ms (features {st.ms_features:.0f} modes {st.ms_modes:.0f} coder {st.ms_coder:.0f})", flush=True)
