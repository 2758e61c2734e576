"""experiment: checkpoint fractions of the pipelined schedule (X3H_PIPE_MARKS) on the dickens-sized stream, config 4's per-GPU share and config 3.
usage: pipe_marks.py ["m1,m2,..." ...]"""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from x3_compressor_amd import _lib, synth
sets = sys.argv[1:] or ["0.02,0.08,0.26,0.62", "0.02,0.08,0.26,0.62,0.85", "0.01,0.04,0.12,0.30,0.60,0.85", "0.02,0.06,0.18,0.40,0.70,0.90", "0.015,0.05,0.14,0.32,0.56,0.78,0.92"]
dev = torch.device("cuda", 0)
CH = 8 << 20
work = {
    "dickens": (synth.english_like(synth.DICKENS_BYTES), np.array([0, synth.DICKENS_BYTES], dtype=np.uint64), dict(w_kib=64, t=256)),
    "config4 share": (synth.zipf_bytes(16 * CH), np.arange(0, 17 * CH, CH, dtype=np.uint64), dict(w_kib=64, t=256)),
}
SIL = [10192446, 51220480, 9970564, 33553445, 6152192, 10085684, 6627202, 21606400, 7251944, 41458703, 8474240, 5345280]
parts = [synth.english_like(n, seed=1000 + i) if i % 3 else synth.zipf_bytes(n, offset=i << 26) for i, n in enumerate(SIL)]
work["config3"] = (np.concatenate(parts), np.cumsum([0] + SIL).astype(np.uint64), dict(w_kib=256, t=1024))
dins = {k: torch.from_numpy(v[0]).to(dev) for k, v in work.items()}
for marks in sets:
    os.environ["X3H_PIPE_MARKS"] = marks
    ctx = _lib.X3Context(0)
    line = [f"marks {marks:40s}"]
    for name, (data, off, kw) in work.items():
        stride = (int((off[1:] - off[:-1]).max()) * 3 // 2 + 4096 + 3) & ~3
        d_out = torch.empty(stride * (len(off) - 1), dtype=torch.uint8, device=dev)
        best = None
        for it in range(3 if name != "config3" else 2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            lens, st = ctx.compress_chunks_dev(dins[name].data_ptr(), off, _lib.make_params(**kw), d_out.data_ptr(), stride)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            if it and (best is None or dt < best[0]): best = (dt, st, int(lens.sum()))
        line.append(f"{name}: {best[0]*1e3:7.1f} ms (coder {best[1].ms_coder:6.1f}, features {best[1].ms_features:5.1f}, out {best[2]})")
        del d_out
    print(" | ".join(line), flush=True)
    ctx.close()
