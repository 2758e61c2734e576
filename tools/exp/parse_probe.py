import sys, os
sys.path.insert(0,'.')
import numpy as np
from x3_compressor_amd import _lib, synth
os.environ["X3H_PIPE_MIN"]="0"
ctx=_lib.X3Context(0, library=sys.argv[1] if len(sys.argv) > 1 else None)
for name,data,kw in (("mr w512 t4096", synth.mr_like(4<<20), dict(w_kib=512,t=4096)),("mr w64 t256", synth.mr_like(4<<20), dict(w_kib=64,t=256)),
                     ("zeros w64 t256", np.zeros(4<<20,np.uint8), dict(w_kib=64,t=256)), ("text w512 t4096", synth.english_like(4<<20), dict(w_kib=512,t=4096)),
                     ("text w64", synth.english_like(4<<20), dict(w_kib=64,t=256))):
    os.environ["X3H_DEBUG"]="1"
    s=ctx.compress(data,_lib.make_params(**kw)); st=ctx.last_stats
    os.environ.pop("X3H_DEBUG")
    print(name, f"parse {st.ms_parse:.1f} ms steps {st.steps} D {st.dict_elems} ratio {data.size/len(s):.2f}", flush=True)
