import sys, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import torch
import oracle_lib
from x3_compressor_amd import _lib, synth
o = oracle_lib.load()
text = synth.english_like(300_000, seed=11)
prm = _lib.make_params(w_kib=8, t=16)
cases = {"a": [0, 100_000, 100_000, 220_000, 300_000], "b": [0, 100_000, 220_000, 300_000], "c": [0, 150_000, 300_000], "d": [0, 300_000]}
which = sys.argv[1]
off = np.array(cases[which], dtype=np.uint64)
with _lib.X3Context(0) as ctx:
    if len(sys.argv) > 2: ctx.set_estimates(True)
    got = ctx.compress_chunks(text, off, prm)
    print(which, 'pipelined', ctx.last_stats.pipelined, [g == o.compress(text[int(off[i]):int(off[i+1])].tobytes(), oracle_lib.params(w_kib=8, t=16)) for i, g in enumerate(got)], flush=True)
