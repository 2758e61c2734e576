"""experiment: single-stream step time, classic vs pipelined schedule, by stream size"""
import sys, os, time
sys.path.insert(0, '.')
import numpy as np, torch
from x3_compressor_amd import _lib, synth
dev = torch.device("cuda", 0)
prm = _lib.make_params(w_kib=64, t=256)
base = synth.english_like(16 << 20)
for mode in ("0", "1"):
    os.environ["X3H_PIPE_MIN"] = mode
    ctx = _lib.X3Context(0)
    for n in (256 << 10, 512 << 10, 1 << 20, 2 << 20, 4 << 20, 16 << 20):
        data = base[:n]
        d_in = torch.from_numpy(np.ascontiguousarray(data)).to(dev)
        stride = (2 * n + 4096 + 3) & ~3
        d_out = torch.empty(stride, dtype=torch.uint8, device=dev)
        off = np.array([0, n], dtype=np.uint64)
        best = 1e9
        for it in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            lens, st = ctx.compress_chunks_dev(d_in.data_ptr(), off, prm, d_out.data_ptr(), stride)
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        print(f"pipe_min={mode} n={n>>10:6d} KiB: {best*1e3:8.2f} ms  {n/best/1e6:6.2f} MB/s  coder {st.ms_coder:.1f} parse {st.ms_parse:.1f} iters {st.mode_iters}", flush=True)
    ctx.close()
