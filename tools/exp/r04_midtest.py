import sys, os
sys.path.insert(0, '.')
import numpy as np
from x3_compressor_amd import _lib, synth
for sizes in ([260_000, 140_000, 99_000, 410_000, 0, 180_000, 300_000, 120_000, 98_304, 650_000], [260_000, 140_000, 99_000, 410_000, 180_000, 300_000, 120_000, 98_304, 650_000], [260_000, 140_000, 180_000, 300_000, 120_000, 650_000]):
    text = synth.english_like(sum(sizes), seed=41)
    off = np.cumsum([0] + sizes).astype(np.uint64)
    data = text.copy()
    if len(sizes) == 10:
        data[int(off[3]):int(off[4])] = synth.zipf_bytes(sizes[3], offset=7 << 20)
        data[int(off[6]):int(off[7])] = synth.mr_like(sizes[6], seed=12)
    prm = _lib.make_params(w_kib=64, t=256)
    ctx = _lib.X3Context(0)
    s = ctx.compress_chunks(data, off, prm)
    print(len(sizes), "streams: pipelined", ctx.last_stats.pipelined, flush=True)
