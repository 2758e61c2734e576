import sys, time; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
from x3_compressor_amd import _lib, synth
nch, cb = 16, 8 << 20
data = synth.zipf_bytes(nch * cb)
off = np.arange(0, (nch + 1) * cb, cb, dtype=np.uint64)
prm = _lib.make_params(w_kib=64, t=256)
ctx = _lib.X3Context(0)
t0 = time.time(); streams = ctx.compress_chunks(data, off, prm, stride=cb + (cb >> 2)); t1 = time.time()
st = ctx.last_stats
tot = sum(map(len, streams))
print(f"batch {nch} x {cb>>20} MiB zipf: wall {t1-t0:.2f}s (incl. H2D/D2H + first allocations), device {st.ms_total:.0f} ms = {nch*cb/st.ms_total/1e3:.1f} MB/s, ratio {nch*cb/tot:.4f}")
print("  stage ms: scan %.1f parse %.1f code %.1f (features %.1f modes %.1f coder %.1f emit %.1f)" % (st.ms_scan, st.ms_parse, st.ms_code, st.ms_features, st.ms_modes, st.ms_coder, st.ms_emit))
t0 = time.time(); streams2 = ctx.compress_chunks(data, off, prm, stride=cb + (cb >> 2)); print(f"  second call wall {time.time()-t0:.2f}s device {ctx.last_stats.ms_total:.0f} ms")
one = ctx.compress(data[3*cb:4*cb], prm)
print("  chunk 3 == standalone stream:", one == streams[3], len(one))
back = ctx.decompress(streams[5], cb)
print("  GPU decode of chunk 5 == input:", back == data[5*cb:6*cb].tobytes())
