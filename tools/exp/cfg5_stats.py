"""stage times of config 5's stream (mr-like samples of the size of Silesia 'mr', -w 512 -t 4096): usage cfg5_stats.py"""
import sys
sys.path.insert(0, '.')
import numpy as np
from x3_compressor_amd import _lib, synth
data = synth.mr_like(9970564, seed=0x5EED)
ctx = _lib.X3Context(0)
prm = _lib.make_params(w_kib=512, t=4096)
for it in range(3):
    s = ctx.compress(data.tobytes(), prm); st = ctx.last_stats
    print(f"total {st.ms_total:.1f} ms: scan {st.ms_scan:.1f} parse {st.ms_parse:.1f} features {st.ms_features:.1f} modes {st.ms_modes:.1f} coder {st.ms_coder:.1f} | steps {st.steps} D {st.dict_elems} symbols {st.chain_symbols} pipelined {st.pipelined} out {len(s)}", flush=True)
