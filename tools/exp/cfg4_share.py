"""one GPU's share of config 4 (16 x 8 MiB Zipf, -w 64 -t 256), device-resident: ms and the pinned chunks"""
import sys, time, json, hashlib, os
sys.path.insert(0, '.')
import numpy as np, torch
from x3_compressor_amd import _lib, synth
CH = 8 << 20
per = int(sys.argv[1]) if len(sys.argv) > 1 else 16
data = synth.zipf_bytes(per * CH)
dev = torch.device("cuda", 0)
d_in = torch.from_numpy(data).to(dev)
stride = (CH + (CH >> 2) + 4096 + 3) & ~3
d_out = torch.empty(stride * per, dtype=torch.uint8, device=dev)
off = np.arange(0, (per + 1) * CH, CH, dtype=np.uint64)
prm = _lib.make_params(w_kib=64, t=256)
man = json.load(open("tests/golden/manifest_sha.json"))
with _lib.X3Context(0) as ctx:
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        lens, st = ctx.compress_chunks_dev(d_in.data_ptr(), off, prm, d_out.data_ptr(), stride)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"{per} x 8 MiB: {dt*1e3:.1f} ms = {per*CH/dt/1e6:.1f} MB/s ratio {per*CH/float(lens.sum()):.4f} pipelined {st.pipelined} parse {st.ms_parse:.1f} features {st.ms_features:.1f} coder {st.ms_coder:.1f} D {st.dict_elems} pairs {st.ctx0_entries}", flush=True)
    ok = {}
    for c in (0, 1, 15):
        if c < per:
            e = man[f"cfg4_zipf_chunk{c}_8m_w64_t256"]
            s = d_out[c * stride:c * stride + int(lens[c])].cpu().numpy().tobytes()
            ok[c] = hashlib.sha256(s).hexdigest() == e["output_sha256"]
    print("pinned chunks", ok)
