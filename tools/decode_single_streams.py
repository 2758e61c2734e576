"""experiment: config 5 (mr-like, -w 512 -t 4096) and the dickens-like stream (-w 64 -t 256): decode time of one stream, chain and second stage apart"""
import os, sys, time
sys.path.insert(0, '.')
from x3_compressor_amd import _lib, synth
ctx = _lib.X3Context(0, library=os.environ.get('X3_LIB'))  # X3_LIB: an experiment build of the library 
for name, data, kw in (("config5 mr-like", synth.mr_like(9970564).tobytes(), dict(w_kib=512, t=4096)),
                       ("dickens-like", synth.english_like(synth.DICKENS_BYTES).tobytes(), dict(w_kib=64, t=256))):
    stream = ctx.compress(data, _lib.make_params(**kw))
    steps = ctx.last_stats.steps
    for rep in range(2):
        t0 = time.time(); back = ctx.decompress(stream, len(data) + 16); dt = time.time() - t0
        st = ctx.last_stats
        print(f"{name}: {len(data)} bytes, {steps} steps; decode wall {dt*1e3:.1f} ms, chain {st.ms_code:.1f} ms = {st.ms_code*1e6/steps:.1f} ns/step, bytes stage {st.ms_emit:.2f} ms, total {st.ms_total:.1f} ms = {len(data)/st.ms_total/1e3:.2f} MB/s, ok {back == data}", flush=True)
