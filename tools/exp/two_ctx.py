"""experiment: the many-chunk batch split over K contexts driven by K host threads on ONE GPU (usage: two_ctx.py [K] [total_MiB] [chunk_KiB])"""
import sys, time, threading
sys.path.insert(0, '.')
import numpy as np, torch
from x3_compressor_amd import _lib, synth
K = int(sys.argv[1]) if len(sys.argv) > 1 else 2
total = (int(sys.argv[2]) if len(sys.argv) > 2 else 256) << 20
cb = (int(sys.argv[3]) if len(sys.argv) > 3 else 256) << 10
base = synth.english_like(8 << 20)
data = np.tile(base, total // base.size)
nch = total // cb
dev = torch.device("cuda", 0)
d_in = torch.from_numpy(data).to(dev)
stride = (cb + (cb >> 1) + 4096 + 3) & ~3
d_out = torch.empty(stride * nch, dtype=torch.uint8, device=dev)
prm = _lib.make_params(w_kib=64, t=256)
ctxs = [_lib.X3Context(0) for _ in range(K)]
per = nch // K
res = [None] * K
def work(k):
    lo, hi = k * per, (k + 1) * per if k < K - 1 else nch
    off = np.arange(lo, hi + 1, dtype=np.uint64) * cb
    res[k] = ctxs[k].compress_chunks_dev(d_in.data_ptr(), off, prm, d_out.data_ptr() + lo * stride, stride)
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(k,)) for k in range(K)]
    [t.start() for t in th]; [t.join() for t in th]
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    tot_out = sum(float(r[0].sum()) for r in res)
    print(f"K={K} run {it}: {nch} chunks x {cb>>10} KiB: wall {dt*1e3:.0f} ms -> {total/dt/1e6:.1f} MB/s ratio {total/tot_out:.3f}", flush=True)
