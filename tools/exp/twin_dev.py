"""Two handles on ONE GPU, each coding half of a many-chunk batch (device-resident in/out) from its own host thread, against one handle coding all of it:
does the coder recurrence of one half (scalar-unit bound, 512 tiny workgroups) hide behind the bandwidth-bound stages of the other?"""
import sys, time, threading
sys.path.insert(0, '.')
import numpy as np, torch
from x3_compressor_amd import _lib, synth
total, cb = 256 << 20, 256 << 10
q = total // 2
data = np.concatenate([synth.english_like(q, seed=0xBA7C4), synth.zipf_bytes(total - q, offset=1 << 33)]).reshape(-1, cb)
order = np.arange(data.shape[0]).reshape(2, -1).T.reshape(-1)          # interleave text and Zipf chunks: both halves get the same mix
data = np.ascontiguousarray(data[order]).reshape(-1)
nch = total // cb
dev = torch.device("cuda", 0)
d_in = torch.from_numpy(data).to(dev)
stride = (cb + (cb >> 1) + 4096 + 3) & ~3
d_out = torch.empty(stride * nch, dtype=torch.uint8, device=dev)
prm = _lib.make_params(w_kib=64, t=256)
ctxs = [_lib.X3Context(0) for _ in range(4)]
def run(k, stagger_ms=0.0):
    per = nch // k
    res = [None] * k
    def work(i):
        if i and stagger_ms: time.sleep(stagger_ms * 1e-3 * i)
        off = np.arange(0, (per + 1) * cb, cb, dtype=np.uint64)
        res[i] = ctxs[i].compress_chunks_dev(d_in.data_ptr() + i * per * cb, off, prm, d_out.data_ptr() + i * per * stride, stride)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(i,)) for i in range(k)]
    [t.start() for t in th]; [t.join() for t in th]
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    return dt, sum(int(r[0].sum()) for r in res)
for k, stg in ((1, 0), (2, 0), (2, 20), (2, 35), (4, 0), (4, 12), (1, 0)):
    for it in range(3):
        dt, out = run(k, stg)
    print(f"{k} handle(s), stagger {stg} ms: wall {dt*1e3:.1f} ms = {total/dt/1e6:.0f} MB/s (compressed {out})", flush=True)
