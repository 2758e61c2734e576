import sys
sys.path.insert(0, '.')
from x3_compressor_amd import _lib, synth
d5 = synth.mr_like(synth.MR_BYTES)
ctx = _lib.X3Context(0)
s = ctx.compress(d5, _lib.make_params(w_kib=512, t=4096))
st = ctx.last_stats
print("steps", st.steps, "parse ms", st.ms_parse, "D", st.dict_elems)
