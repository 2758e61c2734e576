"""experiment: decoding a many-chunk batch (1024 x 256 KiB) in one call"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from x3_compressor_amd import _lib, synth
total, cb = 256 << 20, 256 << 10
base = synth.english_like(8 << 20)
data = np.tile(base, total // base.size)
nch = total // cb
off = np.arange(0, (nch + 1) * cb, cb, dtype=np.uint64)
ctx = _lib.X3Context(0)
prm = _lib.make_params(w_kib=64, t=256)
streams = ctx.compress_chunks(data, off, prm)
print("compressed", sum(map(len, streams)), "D of batch", ctx.last_stats.dict_elems, flush=True)
for it in range(2):
    t0 = time.time(); back = ctx.decompress_chunks(streams, [cb] * nch); dt = time.time() - t0
    st = ctx.last_stats
    print(f"decode run {it}: chain {st.ms_code:.0f} ms + bytes {st.ms_emit:.1f} ms, call {st.ms_total:.0f} ms, wall {dt*1e3:.0f} ms -> {total/(st.ms_code+st.ms_emit)/1e3:.1f} MB/s (both stages)", flush=True)
print("round trip ok:", b"".join(back) == data.tobytes())
