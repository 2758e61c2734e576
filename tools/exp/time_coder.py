"""timing experiment helper: coder time of the dickens-like stream with an alternative build of the library (usage: time_coder.py lib.so)"""
import sys
sys.path.insert(0, '.')
import numpy as np, torch
from x3_compressor_amd import _lib, synth
data = synth.english_like(synth.DICKENS_BYTES)
dev = torch.device("cuda", 0)
d_in = torch.from_numpy(data).to(dev)
stride = (2 * data.size + 4096 + 3) & ~3
d_out = torch.empty(stride, dtype=torch.uint8, device=dev)
off = np.array([0, data.size], dtype=np.uint64)
prm = _lib.make_params(w_kib=64, t=256)
for lib in sys.argv[1:]:
    ctx = _lib.X3Context(0, library=None if lib == "default" else lib)
    for it in range(3):
        import ctypes as C   # experiment builds produce garbage streams (any status): the timings are still valid
        lens, st = np.zeros(1, dtype=np.uint64), _lib.Stats()
        ctx.lib.x3h_compress_chunks_dev(ctx._h, C.byref(prm), C.c_void_p(d_in.data_ptr()), off.ctypes.data, 1, C.c_void_p(d_out.data_ptr()), stride, lens.ctypes.data, C.byref(st))
    print(lib, "ms_total %.1f coder %.1f parse %.1f" % (st.ms_total, st.ms_coder, st.ms_parse), "out", int(lens[0]), flush=True)
    ctx.close()
