"""config 5 (mr-like 16-bit samples, -w 512 -t 4096): where the compress call spends its time (stage sums overlap under the sliced schedule)"""
import sys, time
sys.path.insert(0, '.')
from x3_compressor_amd import _lib, synth
d5 = synth.mr_like(synth.MR_BYTES)
ctx = _lib.X3Context(0)
prm = _lib.make_params(w_kib=512, t=4096)
for rep in range(3):
    t0 = time.time(); s = ctx.compress(d5, prm); dt = time.time() - t0
    st = ctx.last_stats
    print(f"run {rep}: {len(s)} bytes, wall {dt*1e3:.1f} ms, device total {st.ms_total:.1f}: scan {st.ms_scan:.1f} parse {st.ms_parse:.1f} features {st.ms_features:.1f} modes {st.ms_modes:.1f} coder {st.ms_coder:.1f} "
          f"| steps {st.steps} chain symbols {st.chain_symbols} slices {st.coder_launches} schedule {st.pipelined}", flush=True)
