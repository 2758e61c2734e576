"""experiment: where the decoder's cycles go (profile build of decode.hip): usage dec_prof.py lib.so"""
import sys, os, time
sys.path.insert(0, '.')
os.environ["X3H_DEBUG"] = "1"
import numpy as np
from x3_compressor_amd import _lib, synth
ctx = _lib.X3Context(0, library=sys.argv[1] if len(sys.argv) > 1 else None)
prm = _lib.make_params(w_kib=64, t=256)
for n in (160_000, 1_000_000):
    data = synth.english_like(n).tobytes()
    os.environ.pop("X3H_DEBUG", None)
    stream = ctx.compress(data, prm)
    os.environ["X3H_DEBUG"] = "1"
    t0 = time.time(); back = ctx.decompress(stream, n + 16); dt = time.time() - t0
    print(n, back == data, f"{dt*1e3:.1f} ms wall, kernel {ctx.last_stats.ms_code:.1f} ms", flush=True)
