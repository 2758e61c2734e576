"""One stream of the maximum size of Zipf bytes (nearly one parse step per byte: model_events.total reaches 2^28 - 2045, the largest total the library ever divides by): the default
(sliced) schedule against stage-after-stage, byte for byte (--decode: and the stream decoded back on the GPU, minutes).  (Its reference stream is not pinned: the reference needs about a day for it.)"""
import os, sys, time, hashlib
sys.path.insert(0, '.')
import numpy as np
from x3_compressor_amd import _lib, synth
n = (1 << 28) - 4096
data = synth.zipf_bytes(n, offset=3 << 34)
print(f"input: {n} bytes, sha256 {hashlib.sha256(data.tobytes()).hexdigest()}", flush=True)
prm = _lib.make_params(w_kib=1, t=4)
out = {}
for name, env in (("default", {}), ("sequential", {"X3H_PIPE_MIN": "0"})):
    os.environ.pop("X3H_PIPE_MIN", None); os.environ.update(env)
    ctx = _lib.X3Context(0)
    t0 = time.time(); s = ctx.compress(data, prm); dt = time.time() - t0
    st = ctx.last_stats
    out[name] = s
    print(f"{name}: {len(s)} bytes, sha256 {hashlib.sha256(s).hexdigest()}, device {st.ms_total:.0f} ms, wall {dt:.1f} s, steps {st.steps} (event total up to {2051 + st.steps}, 2^28 = {1 << 28}), pipelined {st.pipelined}", flush=True)
    ctx.close()
print("identical:", out["default"] == out["sequential"], flush=True)
if "--decode" not in sys.argv:   # 1.85e8 steps on a dictionary of hundreds of thousands of elements: more than seven minutes of one chain
    sys.exit(0)
os.environ.pop("X3H_PIPE_MIN", None)
ctx = _lib.X3Context(0)
t0 = time.time(); back = ctx.decompress(out["default"], n); dt = time.time() - t0
print(f"decoded {len(back)} bytes in {dt:.0f} s, round trip {'ok' if back == data.tobytes() else 'WRONG'}", flush=True)
