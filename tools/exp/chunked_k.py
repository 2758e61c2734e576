"""experiment: the dickens-sized workload as N chunks, split over K contexts / host threads on ONE GPU (usage: chunked_k.py N K [N K ...])"""
import sys, time, threading
sys.path.insert(0, '.')
import numpy as np, torch
from x3_compressor_amd import _lib, synth
pairs = [(int(sys.argv[i]), int(sys.argv[i + 1])) for i in range(1, len(sys.argv) - 1, 2)] or [(128, 1), (128, 2)]
data = synth.english_like(synth.DICKENS_BYTES)
n = data.size
dev = torch.device("cuda", 0)
d_in = torch.from_numpy(data).to(dev)
prm = _lib.make_params(w_kib=64, t=256)
ctxs = [_lib.X3Context(0) for _ in range(max(k for _, k in pairs))]
for nch, K in pairs:
    cb = (n + nch - 1) // nch
    off = np.array(list(range(0, n, cb)) + [n], dtype=np.uint64)
    nch = len(off) - 1
    stride = (cb + (cb >> 1) + 4096 + 3) & ~3
    d_out = torch.empty(stride * nch, dtype=torch.uint8, device=dev)
    res = [None] * K
    def work(k):
        lo, hi = k * nch // K, (k + 1) * nch // K
        res[k] = ctxs[k].compress_chunks_dev(d_in.data_ptr(), off[lo:hi + 1].copy(), prm, d_out.data_ptr() + lo * stride, stride)
    best = 1e9
    for it in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        th = [threading.Thread(target=work, args=(k,)) for k in range(K)]
        [t.start() for t in th]; [t.join() for t in th]
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        if it: best = min(best, dt)
    tot_out = sum(float(r[0].sum()) for r in res)
    print(f"{nch} chunks, K={K}: wall {best*1e3:.2f} ms -> {n/best/1e6:.1f} MB/s ratio {n/tot_out:.4f}", flush=True)
