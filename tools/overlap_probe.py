"""Does running sub-batches of a many-stream batch side by side (two or three handles on one GPU, started with a stagger so that one handle's per-stream chains meet the
other's HBM-bound stages) beat the one-batch schedule?  usage: overlap_probe.py [total_MiB]"""
import sys, time, threading
sys.path.insert(0, '.')
import numpy as np, torch
from x3_compressor_amd import _lib, synth
total = (int(sys.argv[1]) if len(sys.argv) > 1 else 256) << 20
cb = 256 << 10
data = synth.many_chunks_mix(total)
# interleave text and Zipf chunks so that every sub-batch has the same mix
nch = total // cb
perm = np.arange(nch).reshape(2, nch // 2).T.reshape(-1)
data = data.reshape(nch, cb)[perm].reshape(-1).copy()
dev = torch.device("cuda", 0)
d_in = torch.from_numpy(data).to(dev)
stride = (cb + (cb >> 1) + 4096 + 3) & ~3
d_out = torch.empty(stride * nch, dtype=torch.uint8, device=dev)
prm = _lib.make_params(w_kib=64, t=256)
ctxs = [_lib.X3Context(0) for _ in range(4)]

def run_part(ctx, lo, hi, res, k):
    off = np.arange(0, (hi - lo + 1) * cb, cb, dtype=np.uint64)
    lens, st = ctx.compress_chunks_dev(d_in.data_ptr() + lo * cb, off, prm, d_out.data_ptr() + lo * stride, stride)
    res[k] = int(lens.sum())

def timed(parts, lanes, stagger_ms):
    """parts: list of (lo, hi); lane l runs parts l, l + lanes, ... one after the other; lane l starts l * stagger_ms late"""
    res = [0] * len(parts)
    def lane(l):
        time.sleep(l * stagger_ms * 1e-3)
        for k in range(l, len(parts), lanes):
            run_part(ctxs[l], parts[k][0], parts[k][1], res, k)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    th = [threading.Thread(target=lane, args=(l,)) for l in range(lanes)]
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3, sum(res)

def split(k):
    return [(i * nch // k, (i + 1) * nch // k) for i in range(k)]

for lanes, k, stag in [(1, 1, 0), (1, 1, 0), (1, 2, 0), (1, 4, 0), (2, 2, 0), (2, 2, 0), (2, 2, 8), (2, 2, 16), (2, 2, 24), (2, 4, 0), (2, 4, 6), (2, 4, 12), (3, 6, 6), (3, 3, 10), (4, 4, 8), (4, 8, 4), (2, 8, 4), (1, 1, 0)]:
    best = None
    for rep in range(3):
        ms, tot = timed(split(k), lanes, stag)
        if best is None or ms < best: best = ms
    print(f"{k} sub-batches on {lanes} handle(s), stagger {stag:2d} ms: best of 3 {best:6.1f} ms = {total / best / 1e3:7.1f} MB/s (compressed {tot})", flush=True)
