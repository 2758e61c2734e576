"""All five BASELINE.json configs at full size on one GPU (synthetic stand-ins for Silesia; config 4 = one GPU's share of the 128 chunks).
Prints one line per config: bytes, device ms, MB/s, ratio, and the checks made.
usage: configs_check.py [--corpus DIR [--manifest tests/golden/manifest_corpus.json]]
       --corpus DIR: ALSO run the real Silesia files found in DIR (configs 2, 3 and 5 name them) and compare every stream's sha256 with the manifest
       that tests/golden/make_golden_sha.py --corpus DIR wrote with the real reference (files without an entry are reported as "not pinned")."""
import hashlib, json, os, sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from x3_compressor_amd import _lib, synth
import oracle_lib

ctx = _lib.X3Context(0)
SILESIA = synth.SILESIA
MAN = json.load(open('tests/golden/manifest_sha.json'))  # sha256 of the REAL reference's streams (tests/golden/make_golden_sha.py)


def pinned(name, stream):
    e = MAN.get(name)
    return None if e is None else (len(stream) == e["output_len"] and hashlib.sha256(stream).hexdigest() == e["output_sha256"])


def line(name, nbytes, st, out_bytes, extra=""):
    print(f"{name:58s} {nbytes/1e6:8.1f} MB  {st.ms_total:9.1f} ms  {nbytes/st.ms_total/1e3:8.1f} MB/s  ratio {nbytes/out_bytes:6.3f}  {extra}", flush=True)


# config 1: first 64 KiB, -w 8 -t 16, GPU stream == CPU oracle stream
d = synth.english_like(65536).tobytes()
s = ctx.compress(d, _lib.make_params(w_kib=8, t=16)); st = ctx.last_stats
ok = s == oracle_lib.load().compress(d, oracle_lib.params(w_kib=8, t=16))
line("1: 64 KiB text, -w 8 -t 16", len(d), st, len(s), f"== CPU oracle stream: {ok}")

# config 2: dickens-sized stream, -w 64 -t 256 (the bench.py workload)
d = synth.english_like(SILESIA["dickens"]).tobytes()
for _ in range(2):
    s = ctx.compress(d, _lib.make_params(w_kib=64, t=256)); st = ctx.last_stats
line("2: 10.2 MB text, -w 64 -t 256 (host buffers)", len(d), st, len(s), f"sha256 == real reference: {pinned('cfg2_full_english10192446_w64_t256', s)}")

# config 3: 12 independent streams with the Silesia sizes, -w 256 -t 1024
sizes = list(SILESIA.values())
parts = [synth.config3_part(i) for i in range(len(sizes))]
data = np.concatenate(parts); off = np.cumsum([0] + sizes).astype(np.uint64)
for _ in range(2):
    streams = ctx.compress_chunks(data, off, _lib.make_params(w_kib=256, t=1024)); st = ctx.last_stats
oks = [pinned(f"cfg3_full_{i:02d}_{nm}{sizes[i]}_w256_t1024", streams[i]) for i, nm in enumerate(SILESIA)]
line("3: 12 streams with the Silesia sizes, -w 256 -t 1024", int(off[-1]), st, sum(map(len, streams)), f"pipelined {st.pipelined}; streams with sha256 == real reference: {sum(1 for o in oks if o)} of 12 (differ: {sum(1 for o in oks if o is False)})")

# config 4 (one GPU's share): 16 chunks x 8 MiB of the Zipf stream, -w 64 -t 256
nch, cb = 16, 8 << 20
data = synth.zipf_bytes(nch * cb); off = np.arange(0, (nch + 1) * cb, cb, dtype=np.uint64)
for _ in range(2):
    streams = ctx.compress_chunks(data, off, _lib.make_params(w_kib=64, t=256), stride=cb + (cb >> 2)); st = ctx.last_stats
line("4: 16 x 8 MiB Zipf (1/8 of the 1 GiB), -w 64 -t 256", nch * cb, st, sum(map(len, streams)), f"pipelined {st.pipelined}")

# config 5: mr-sized stream, -w 512 -t 4096, compress + decompress round trip
d = synth.mr_like(synth.MR_BYTES).tobytes()
for _ in range(2):
    s = ctx.compress(d, _lib.make_params(w_kib=512, t=4096)); st = ctx.last_stats
t0 = time.time(); back = ctx.decompress(s, len(d)); dt = time.time() - t0
line("5: 10.0 MB of mr-like samples, -w 512 -t 4096, round trip", len(d), st, len(s), f"sha256 == real reference: {pinned(f'cfg5_full_mr{len(d)}_w512_t4096', s)}; decode {ctx.last_stats.ms_code:.0f} ms ({len(d)/dt/1e6:.2f} MB/s), round trip ok: {back == d}")


# the real corpus, if a directory was given: the same three configs on the files themselves
if "--corpus" in sys.argv:
    sys.path.insert(0, 'tests/golden')
    import make_golden_sha
    cdir = sys.argv[sys.argv.index("--corpus") + 1]
    mpath = sys.argv[sys.argv.index("--manifest") + 1] if "--manifest" in sys.argv else 'tests/golden/manifest_corpus.json'
    cman = json.load(open(mpath)) if os.path.exists(mpath) else {}
    bad = 0
    for name, (path, args) in sorted(make_golden_sha.corpus_cases(cdir).items()):
        d = open(path, "rb").read()
        if len(d) > (128 << 20):
            print(f"{name}: {len(d)} bytes exceed one stream (X3H_MAX_CHUNK), skipped"); continue
        w, t = int(args[args.index("-w") + 1]), int(args[args.index("-t") + 1])
        s = ctx.compress(d, _lib.make_params(w_kib=w, t=t)); st = ctx.last_stats
        e = cman.get(name)
        ok = None if e is None or e["input_sha256"] != hashlib.sha256(d).hexdigest() else (len(s) == e["output_len"] and hashlib.sha256(s).hexdigest() == e["output_sha256"])
        bad += ok is False
        line(name, len(d), st, len(s), "sha256 == real reference: " + ("not pinned (run tests/golden/make_golden_sha.py --corpus)" if ok is None else str(ok)))
    sys.exit(1 if bad else 0)
