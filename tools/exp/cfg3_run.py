"""config 3 (12 streams with the Silesia sizes, -w 256 -t 1024) once per schedule, device-resident; X3H_DEBUG=1 prints the time line"""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from x3_compressor_amd import _lib, synth
sizes = list(synth.SILESIA.values())
parts = [synth.config3_part(i) for i in range(len(sizes))]
data = np.concatenate(parts); off = np.cumsum([0] + sizes).astype(np.uint64)
dev = torch.device("cuda", 0)
d_in = torch.from_numpy(data).to(dev)
stride = (max(sizes) + (max(sizes) >> 2) + 4096 + 3) & ~3
d_out = torch.empty(stride * len(sizes), dtype=torch.uint8, device=dev)
prm = _lib.make_params(w_kib=256, t=1024)
with _lib.X3Context(0) as ctx:
    for it in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        lens, st = ctx.compress_chunks_dev(d_in.data_ptr(), off, prm, d_out.data_ptr(), stride)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"config 3: {dt*1e3:.1f} ms = {int(off[-1])/dt/1e6:.1f} MB/s pipelined {st.pipelined} parse {st.ms_parse:.1f} features {st.ms_features:.1f} coder {st.ms_coder:.1f} symbols {st.chain_symbols} D {st.dict_elems}", flush=True)
