"""Randomised differential test on the GPU: libx3hip.so vs the CPU oracle over random inputs, parameters, batch shapes and schedules.
usage: fuzz_parity.py [seconds] [seed]   (prints the first mismatch with everything needed to reproduce it; exit code 1)"""
import os, sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import oracle_lib
from x3_compressor_amd import _lib, synth

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 12345
rng = np.random.default_rng(seed)
orc = oracle_lib.load()
TEXT = synth.english_like(1 << 20, seed=seed).tobytes()
ZIPF = synth.zipf_bytes(1 << 20, offset=seed & 0xFFFF).tobytes()


def gen(n):
    kind = int(rng.integers(0, 8))
    if n == 0: return b""
    o = int(rng.integers(0, (1 << 20) - n)) if n < (1 << 20) else 0
    if kind == 0: return TEXT[o:o + n]
    if kind == 1: return ZIPF[o:o + n]
    if kind == 2: return bytes(rng.integers(0, 256, n, dtype=np.uint8))
    if kind == 3: return bytes(rng.integers(0, int(rng.integers(1, 5)), n, dtype=np.uint8))          # tiny alphabet, zero bytes included
    if kind == 4: p = bytes(rng.integers(0, 256, int(rng.integers(1, 40)), dtype=np.uint8)); return (p * (n // len(p) + 1))[:n]   # periodic
    if kind == 5: return TEXT[o:o + n // 2] + bytes(n - n // 2)                                         # text then a run of zeros
    if kind == 6: return synth.mr_like(n, seed=int(rng.integers(0, 1 << 30))).tobytes()                  # sparse 16-bit samples: dense classes
    a = np.zeros(n, np.uint8); k = max(1, n // 12); a[rng.integers(0, n, k)] = rng.integers(1, 4, k); return a.tobytes()   # zeros with sparse noise


ctxs = {}
def ctx_for(env):
    key = tuple(sorted(env.items()))
    if key not in ctxs:
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        ctxs[key] = _lib.X3Context(0)
        for k, v in old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v
    return ctxs[key]


ENVS = [dict(), dict(X3H_PIPE_MIN="1"), dict(X3H_PIPE_MIN="1", X3H_MODES="fixed"), dict(X3H_PIPE_MIN="0", X3H_MODES="serial"), dict(X3H_PIPE_MIN="0", X3H_MODES="fixed"),
        dict(X3H_STREAM_KERNELS="1", X3H_PIPE_MIN="0"), dict(X3H_STREAM_KERNELS="0"), dict(X3H_WALK_DENSE="16"), dict(X3H_WALK_DENSE="40", X3H_STREAM_KERNELS="1", X3H_PIPE_MIN="0"),
        dict(X3H_PIPE_MIN="1", X3H_MODES="serial"), dict(X3H_CTX_SUB="3"), dict(X3H_PIPE_MIN="1", X3H_PIPE_MARKS="0.1,0.2,0.3,0.4,0.5,0.6,0.7,0.8"),
        dict(X3H_PIPE_MIN="1", X3H_PIPE_MARKS="0.6"),
        # round 3: K1 by one workgroup per chunk (scan3.hip) on batches of any size, in both forms, with the per-chunk / chip-wide dense refinement;
        # segment-wise bit emission; per-stream arrangement
        dict(X3H_SEG_MIN="1"), dict(X3H_SEG_MIN="1", X3H_SEG_SMALL_MAX="0"), dict(X3H_SEG_MIN="1", X3H_WALK_DENSE="12"), dict(X3H_SEG_MIN="1", X3H_SEG_REFINE="0", X3H_WALK_DENSE="12"),
        dict(X3H_SEG_MIN="2", X3H_STREAM_KERNELS="1", X3H_PIPE_MIN="0"), dict(X3H_PIPE_MIN="1", X3H_SEG_EMIT="1"), dict(X3H_STREAM_KERNELS="1", X3H_PIPE_MIN="0", X3H_ARRANGE="1"),
        # round 3, second half: coder chains four to a workgroup (default from 513 streams), context kernel with / without its own tag gather and XCD mapping,
        # move-to-front ranks by one wavefront per stream instead of eight time ranges
        dict(X3H_AC2_WIDE="1", X3H_STREAM_KERNELS="1", X3H_PIPE_MIN="0"), dict(X3H_CTX_GATHER="0"), dict(X3H_CTX_SUB="64", X3H_CTX_XCD="0"), dict(X3H_CTX_SUB="1", X3H_CTX_XCD="1"),
        dict(X3H_MTF_PAR="0", X3H_IDX_PAR="0"), dict(X3H_MTF_PAR="0", X3H_STREAM_KERNELS="1", X3H_PIPE_MIN="0", X3H_CTX_GATHER="0"), dict(X3H_IDX_PAR="0", X3H_STREAM_KERNELS="1", X3H_PIPE_MIN="0"),
        # round 4: K3 in slices forced on small inputs (many small slices / a few / stage B on the parse stream / several wavefronts per stream's contexts), and switched off
        dict(X3H_SLICED_MIN="1", X3H_SLICE_GAP="64", X3H_SLICE_MARKS="0.02,0.05,0.10,0.17,0.26,0.36,0.47,0.59,0.72,0.86"),
        dict(X3H_SLICED_MIN="1", X3H_SLICE_GAP="300", X3H_SLICE_MARKS="0.1,0.3,0.6", X3H_SLICE_SUB="3"),
        dict(X3H_SLICED_MIN="1", X3H_SLICE_GAP="128", X3H_SLICE_MARKS="0.02,0.05,0.10,0.17,0.26,0.36,0.47,0.59,0.72,0.86", X3H_SLICE_BSTREAM="1", X3H_SLICE_SUB="7"),
        dict(X3H_SLICED_MIN="1", X3H_SLICE_GAP="1000"), dict(X3H_SLICED="0", X3H_PIPE_MIN="1"),
        # round 4, second half: the per-stream sort of the context arrangements (x3_segsort_kernel: default from 128 streams on) forced on any batch, in its forms;
        # K1's level tests reading their list entries from memory instead of the LDS copy
        dict(X3H_SEGSORT="1", X3H_STREAM_KERNELS="1", X3H_PIPE_MIN="0"), dict(X3H_SEGSORT="1", X3H_SEGSORT_NINE="1", X3H_STREAM_KERNELS="1", X3H_PIPE_MIN="0"),
        dict(X3H_SEGSORT="1", X3H_SEGSORT_PASSES="3", X3H_SEGSORT_GEN="1", X3H_STREAM_KERNELS="1", X3H_PIPE_MIN="0"),
        dict(X3H_SEGSORT="1", X3H_SEGSORT_DBITS8="1", X3H_SEGSORT_PASSES="2", X3H_STREAM_KERNELS="1", X3H_PIPE_MIN="0", X3H_SEG_MIN="1"),
        dict(X3H_SEG_MIN="1", X3H_SEG_LA_LDS="0"), dict(X3H_SEG_MIN="1", X3H_SEG_LA_LDS="0", X3H_SEG_SMALL_MAX="0"),
        # round 5: slices arranged by the per-stream sort (x3_segsort_kernel: what slices above 32 768 hits per stream take) instead of the LDS counting sort; every
        # environment runs the hand-written radix sort / chained scans of prims.hip (K1 of few streams, the generic coding stage) and K2's prefix-hash block fill
        dict(X3H_SLICED_MIN="1", X3H_SLICE_GAP="64", X3H_SLICE_ARRANGE="0"), dict(X3H_SLICED_MIN="1", X3H_SLICE_GAP="300", X3H_SLICE_ARRANGE="0", X3H_SLICE_SEGSORT="1", X3H_SLICE_SUB="3"),
        dict(X3H_SLICED_MIN="1", X3H_SLICE_GAP="1000", X3H_SLICE_ARRANGE="0", X3H_SLICE_MARKS="0.004,0.02,0.05,0.10,0.17,0.26,0.36,0.47,0.59,0.72,0.86")]
t0, cases = time.time(), 0
while time.time() - t0 < budget:
    nch = int(rng.choice([1, 1, 1, 2, 3, 7, 40, 60, 130, 300, 600]))   # >= 48: the per-stream kernels of code3.hip; 300: above the switch to the small-LDS kernel variants
    sizes = [int(rng.integers(0, 3000)) for _ in range(nch)] if nch > 100 else [int(rng.integers(0, 9000)) for _ in range(nch)] if nch > 45 else [int(rng.choice([0, 1, 2, 31, 32, 33, 200, 2047, 2048, 2049, 5000, 20000, 70000])) if rng.random() < 0.5 else int(rng.integers(0, 30000)) for _ in range(nch)]
    w = int(rng.choice([0, 1, 1, 2, 4, 8, 8, 16, 64]))
    kw = dict(w_kib=w, t=int(rng.choice([0, 1, 2, 3, 8, 15, 16, 64, 256, 5000])), m=int(rng.choice([0, 1, 4, 4, 4, 9])),
              n=int(rng.choice([0, 0, 0, 1, 2, 5])), x=int(rng.random() < 0.15))
    if w == 0: continue  # -w 0 reads out of bounds in the reference itself (SURVEY 8a trap 10)
    parts = [gen(n) for n in sizes]
    env = ENVS[int(rng.integers(0, len(ENVS)))]
    ctx = ctx_for(env)
    prm, oprm = _lib.make_params(**kw), oracle_lib.params(**kw)
    want = [orc.compress(p, oprm) for p in parts]
    data = np.frombuffer(b"".join(parts), dtype=np.uint8)
    off = np.cumsum([0] + sizes).astype(np.uint64)
    got = ctx.compress_chunks(data, off, prm) if nch > 1 or rng.random() < 0.3 else [ctx.compress(parts[0], prm)]
    cases += 1
    for i in range(nch):
        if got[i] != want[i]:
            print("MISMATCH: seed", seed, "case", cases, "chunk", i, "sizes", sizes, "params", kw, "env", env, flush=True)
            sys.exit(1)
    if rng.random() < 0.5:   # decoder: the whole batch back (more than 256 streams: the small-LDS kernel variant, whose tables spill at 512 elements)
        back = ctx.decompress_chunks(got, [n + int(rng.integers(0, 9)) for n in sizes])
        for i in range(nch):
            if back[i] != parts[i]:
                print("BATCH DECODE MISMATCH: seed", seed, "case", cases, "chunk", i, "sizes", sizes, "params", kw, "env", env, flush=True)
                sys.exit(1)
    if rng.random() < 0.3:   # decoder round trip of one stream
        i = int(rng.integers(0, nch))
        back = ctx.decompress(got[i], sizes[i] + 8)
        if back != parts[i]:
            print("DECODE MISMATCH: seed", seed, "case", cases, "chunk", i, "sizes", sizes, "params", kw, "env", env, flush=True)
            sys.exit(1)
    if cases % 25 == 0:
        print(f"{cases} cases ok, {time.time() - t0:.0f} s", flush=True)
print(f"fuzz ok: {cases} cases in {time.time() - t0:.0f} s (seed {seed})")
