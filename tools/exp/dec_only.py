"""experiment: one batch decode (64 streams of the dickens-like text) for counter collection"""
import sys
sys.path.insert(0, '.')
import numpy as np
from x3_compressor_amd import _lib, synth
ctx = _lib.X3Context(0)
prm = _lib.make_params(w_kib=64, t=256)
data = synth.english_like(synth.DICKENS_BYTES)
cb = (data.size + 63) // 64
off = np.array(list(range(0, data.size, cb)) + [data.size], dtype=np.uint64)
streams = ctx.compress_chunks(data, off, prm)
caps = [int(off[i + 1] - off[i]) for i in range(len(off) - 1)]
back = ctx.decompress_chunks(streams, caps)
print("ok", b"".join(back) == data.tobytes(), ctx.last_stats.ms_code)
