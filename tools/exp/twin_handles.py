"""Do two (or four) handles on ONE GPU, each coding half (a quarter) of a many-chunk batch from its own host thread, beat one handle coding all of it?
(the coder recurrence is bound by the scalar units, most other kernels by memory: kernels of different sub-batches can share the chip)"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from x3_compressor_amd import _lib, synth
total, cb = 256 << 20, 256 << 10
q = total // 2
data = np.concatenate([synth.english_like(q, seed=0xBA7C4), synth.zipf_bytes(total - q, offset=1 << 33)])
# interleave text and Zipf chunks so that every handle gets the same mix
data = data.reshape(-1, cb)
order = np.arange(data.shape[0]).reshape(2, -1).T.reshape(-1)
data = np.ascontiguousarray(data[order]).reshape(-1)
off = np.arange(0, total + 1, cb, dtype=np.uint64)
prm = _lib.make_params(w_kib=64, t=256)
handles = [_lib.X3Context(0) for _ in range(4)]
ref = None
for nh in (1, 2, 4, 2, 1):
    for it in range(3):
        t0 = time.perf_counter()
        got = _lib.compress_chunks_multi(handles[:nh], data, off, prm)
        dt = time.perf_counter() - t0
        st = handles[0].last_stats
        print(f"{nh} handle(s), run {it}: wall {dt*1e3:.0f} ms = {total/dt/1e6:.0f} MB/s (device ms_total of the slowest handle {st.ms_total:.0f})", flush=True)
    if ref is None: ref = got
    else: assert got == ref
