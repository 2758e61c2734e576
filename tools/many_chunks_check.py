"""Aggregate throughput of one batch of many independent chunks (device-resident in/out).
usage: many_chunks_check.py [total_MiB] [chunk_KiB] [text|mix|mr]   (mix = bench.py's fresh batch: 1/2 text, 1/2 Zipf bytes; mr = mr-like samples)"""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from x3_compressor_amd import _lib, synth
total = (int(sys.argv[1]) if len(sys.argv) > 1 else 256) << 20
cb = (int(sys.argv[2]) if len(sys.argv) > 2 else 256) << 10
if len(sys.argv) > 3 and sys.argv[3] == "mix":
    q = total // 2
    data = np.concatenate([synth.english_like(q, seed=0xBA7C4), synth.zipf_bytes(total - q, offset=1 << 33)])
elif len(sys.argv) > 3 and sys.argv[3] == "mr":
    data = synth.mr_like(total, seed=0xBA7)
else:
    base = synth.english_like(8 << 20)
    data = np.tile(base, total // base.size)        # chunks are independent streams, so repeated content costs the same as fresh content
nch = total // cb
off = np.arange(0, (nch + 1) * cb, cb, dtype=np.uint64)
dev = torch.device("cuda", 0)
d_in = torch.from_numpy(data).to(dev)
stride = (cb + (cb >> 1) + 4096 + 3) & ~3
d_out = torch.empty(stride * nch, dtype=torch.uint8, device=dev)
ctx = _lib.X3Context(0)
import os
prm = _lib.make_params(w_kib=int(os.environ.get("MC_W", "64")), t=int(os.environ.get("MC_T", "256")))
for it in range(int(os.environ.get('MC_RUNS', '3'))):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    lens, st = ctx.compress_chunks_dev(d_in.data_ptr(), off, prm, d_out.data_ptr(), stride)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"run {it}: {nch} chunks x {cb>>10} KiB = {total>>20} MiB: wall {dt*1e3:.1f} ms -> {total/dt/1e6:.1f} MB/s | device ms: scan {st.ms_scan:.1f} parse {st.ms_parse:.1f} code {st.ms_code:.1f} (features {st.ms_features:.1f} modes {st.ms_modes:.1f} coder {st.ms_coder:.1f} emit {st.ms_emit:.1f}) ratio {total/float(lens.sum()):.3f}", flush=True)
print("mem GB", torch.cuda.mem_get_info()[0] / 1e9, "free of", torch.cuda.mem_get_info()[1] / 1e9)
