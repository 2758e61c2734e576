"""One stream of the maximum chunk size (X3H_MAX_CHUNK = 128 MiB): pipelined schedule vs stage-after-stage, byte for byte."""
import os, sys, time, hashlib
sys.path.insert(0, '.')
import numpy as np
from x3_compressor_amd import _lib, synth
n = 128 << 20
data = np.concatenate([synth.english_like(64 << 20, seed=3), synth.zipf_bytes(64 << 20)]).tobytes()
prm = _lib.make_params(w_kib=64, t=256)
out = {}
for name, env in (("pipelined", {}), ("sequential", {"X3H_PIPE_MIN": "0"})):
    os.environ.pop("X3H_PIPE_MIN", None); os.environ.update(env)
    ctx = _lib.X3Context(0)
    ctx.compress(data, prm)  # first call: workspace allocation
    t0 = time.time(); s = ctx.compress(data, prm); dt = time.time() - t0
    st = ctx.last_stats
    out[name] = hashlib.sha256(s).hexdigest()
    print(f"{name}: {len(s)} bytes, device {st.ms_total:.0f} ms ({n/st.ms_total/1e3:.1f} MB/s), wall {dt:.1f} s, steps {st.steps}, symbols {st.coded_symbols}, pipelined {st.pipelined}", flush=True)
    ctx.close()
print("identical:", out["pipelined"] == out["sequential"])
