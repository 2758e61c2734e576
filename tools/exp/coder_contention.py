"""experiment: coder-chain time per symbol against the number of chains per CU (256 MiB of text as N chunks).  usage: coder_contention.py"""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from x3_compressor_amd import _lib, synth
total = 256 << 20
base = synth.english_like(8 << 20)
data = np.tile(base, total // base.size)
dev = torch.device("cuda", 0)
d_in = torch.from_numpy(data).to(dev)
ctx = _lib.X3Context(0)
prm = _lib.make_params(w_kib=64, t=256)
for nch in (128, 256, 512, 1024, 2048, 4096):
    cb = total // nch
    off = np.arange(0, (nch + 1) * cb, cb, dtype=np.uint64)
    stride = (cb + (cb >> 1) + 4096 + 3) & ~3
    d_out = torch.empty(stride * nch, dtype=torch.uint8, device=dev)
    for it in range(2):
        lens, st = ctx.compress_chunks_dev(d_in.data_ptr(), off, prm, d_out.data_ptr(), stride)
    per_chain = st.chain_symbols / nch
    print(f"{nch:5d} chains ({nch/256:.1f} per CU), {per_chain:9.0f} symbols each: coder {st.ms_coder:7.2f} ms = {st.ms_coder*1e6/per_chain:6.1f} ns per symbol per chain; pipelined {st.pipelined}", flush=True)
    del d_out
