"""Decoder throughput of one batch of independent chunk streams, by chunk size (one wavefront per stream: the batch rate is
streams in flight x the per-stream rate).  usage: decode_batch_check.py [total_MiB] [chunk_KiB ...]"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from x3_compressor_amd import _lib, synth
total = (int(sys.argv[1]) if len(sys.argv) > 1 else 256) << 20
sizes = [int(a) for a in sys.argv[2:]] or [256, 128, 64, 32]
base = synth.english_like(8 << 20)
data = np.tile(base, max(total // base.size, 1))[:total]
import os
ctx = _lib.X3Context(0, library=os.environ.get("X3_LIB"))
prm = _lib.make_params(w_kib=64, t=256)
for kib in sizes:
    cb = kib << 10
    nch = total // cb
    off = np.arange(0, (nch + 1) * cb, cb, dtype=np.uint64)
    streams = ctx.compress_chunks(data, off, prm)
    comp = sum(map(len, streams))
    for it in range(2):
        t0 = time.perf_counter()
        back = ctx.decompress_chunks(streams, [cb + 8] * nch)
        dt = time.perf_counter() - t0
        st = ctx.last_stats
        print(f"{nch} streams x {kib} KiB (ratio {total / comp:.3f}): decode kernel {st.ms_code:.0f} ms = {total / st.ms_code / 1e3:.0f} MB/s, "
              f"library call incl. H2D/D2H {st.ms_total:.0f} ms, wall {dt * 1e3:.0f} ms", flush=True)
    assert b"".join(back) == data.tobytes()
print("round trips ok")
