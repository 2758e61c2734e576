"""One stream of the maximum chunk size (X3H_MAX_CHUNK = 2^28 - 4096 bytes): the default (sliced) schedule against stage-after-stage, byte for byte, and a decode of a
cheap stream of that size (zeros).  Prints the stream's sha256 (tests/golden/manifest_sha.json: big_english_maxchunk_w1_t4 is the real reference's)."""
import os, sys, time, hashlib
sys.path.insert(0, '.')
import numpy as np
from x3_compressor_amd import _lib, synth
n = (1 << 28) - 4096
t0 = time.time()
data = synth.english_like(n, seed=55)
print(f"input: {n} bytes, sha256 {hashlib.sha256(data.tobytes()).hexdigest()} ({time.time() - t0:.0f} s to generate)", flush=True)
prm = _lib.make_params(w_kib=1, t=4)
out = {}
for name, env in (("default", {}), ("sequential", {"X3H_PIPE_MIN": "0"})):
    os.environ.pop("X3H_PIPE_MIN", None); os.environ.update(env)
    ctx = _lib.X3Context(0)
    t0 = time.time(); s = ctx.compress(data, prm); dt = time.time() - t0
    st = ctx.last_stats
    out[name] = hashlib.sha256(s).hexdigest()
    print(f"{name}: {len(s)} bytes, sha256 {out[name]}, device {st.ms_total:.0f} ms ({n/st.ms_total/1e3:.1f} MB/s), wall {dt:.1f} s, steps {st.steps}, symbols {st.coded_symbols}, pipelined {st.pipelined}", flush=True)
    ctx.close()
print("identical:", out["default"] == out["sequential"], flush=True)
os.environ.pop("X3H_PIPE_MIN", None)
ctx = _lib.X3Context(0)
z = np.zeros(n, dtype=np.uint8)
zs = ctx.compress(z, _lib.make_params(w_kib=8, t=16))
t0 = time.time(); back = ctx.decompress(zs, n); dt = time.time() - t0
print(f"zeros: {len(zs)}-byte stream, decoded {len(back)} bytes in {dt:.1f} s, round trip {'ok' if back == z.tobytes() else 'WRONG'}", flush=True)
