"""The HIP kernel SOURCES (scan.hip / parse.hip / code.hip / api.hip) executed on the CPU by the SIMT emulator of
tests/emu (functional model: fibers + lockstep wave intrinsics) against the oracle, on tiny inputs.  This is a logic
check that runs without a GPU; the parity tests proper are tests/test_gpu_parity.py (-m gpu) on the real library."""
import os
import subprocess

import numpy as np
import pytest

import oracle_lib
from x3_compressor_amd import _lib, synth

HERE = os.path.dirname(os.path.abspath(__file__))
EMU_SO = os.path.join(HERE, "emu", "libx3emu.so")


@pytest.fixture(scope="module")
def emu():
    subprocess.run(["make", "-C", os.path.join(HERE, "emu")], check=True, capture_output=True)
    ctx = _lib.X3Context(0, library=EMU_SO)
    yield ctx
    ctx.close()


CASES = [
    ("empty", b"", dict()),
    ("one", b"a", dict()),
    ("abra", b"abracadabra" * 30, dict(w_kib=1, t=2)),
    ("english_refill", synth.english_like(2600).tobytes(), dict(w_kib=1, t=4)),   # crosses the 2048-position LDS block
    ("zipf", synth.zipf_bytes(1500).tobytes(), dict(w_kib=1, t=2)),
    ("zeros", bytes(1200), dict(w_kib=1, t=15)),
    ("factor2", synth.english_like(1200).tobytes(), dict(w_kib=1, t=3, n=2)),
    ("nl_mode", synth.english_like(1200).tobytes(), dict(w_kib=1, t=3, x=1)),
]


@pytest.mark.parametrize("name,data,kw", CASES, ids=[c[0] for c in CASES])
def test_emulated_kernels_match_oracle(emu, oracle, name, data, kw):
    prm, oprm = _lib.make_params(**kw), oracle_lib.params(**kw)
    stream, opos, oinfo, st = oracle.trace(data, oprm)
    tp, ti, d = emu.parse(data, prm)
    assert np.array_equal(tp, opos) and np.array_equal(ti, oinfo) and d == st.dict_elems
    assert emu.compress(data, prm) == stream


SEG_CASES = [
    ("abra", b"abracadabra" * 30, dict(w_kib=1, t=2)),
    ("english", synth.english_like(2600).tobytes(), dict(w_kib=1, t=4)),
    ("zipf_rare_bytes", synth.zipf_bytes(1500).tobytes(), dict(w_kib=1, t=2)),          # most bytes occur fewer than T+1 times in a window: exact K
    ("zeros_dense", bytes(1200), dict(w_kib=1, t=15)),                                 # one dense class + the padding zeros as occurrences
    ("mr_like", synth.mr_like(3000).tobytes(), dict(w_kib=1, t=3)),
    ("two_tiles_t1", synth.english_like(5000, seed=5).tobytes(), dict(w_kib=2, t=1)),   # more than one 4096-element tile per pass
    ("one_byte", b"a", dict()),
]


@pytest.mark.parametrize("name,data,kw,form", [c + ("counters-in-lds",) for c in SEG_CASES] + [c + ("counters-in-global-memory",) for c in SEG_CASES[1:4]],
                         ids=[c[0] + "-lds" for c in SEG_CASES] + [c[0] + "-global" for c in SEG_CASES[1:4]])
def test_emulated_per_chunk_scan_matches_oracle(emu, oracle, monkeypatch, name, data, kw, form):
    """scan3.hip (K1 of many-chunk batches: one workgroup sorts and level-tests one chunk) forced on a single chunk, in both of its forms
    (chunks up to 256 KiB / longer ones): m[] == the oracle's (backend.c:56-78) and the stream built on it == the oracle's"""
    monkeypatch.setenv("X3H_SEG_MIN", "1")
    if form == "counters-in-global-memory":
        monkeypatch.setenv("X3H_SEG_SMALL_MAX", "0")
    prm, oprm = _lib.make_params(**kw), oracle_lib.params(**kw)
    assert np.array_equal(np.frombuffer(emu.scan_m(data, prm), np.uint8), np.frombuffer(oracle.scan_m(data, oprm), np.uint8))
    assert emu.compress(data, prm) == oracle.compress(data, oprm)


def test_emulated_per_chunk_scan_batch(emu, oracle, monkeypatch):
    """a ragged batch (empty chunk, one byte, text, zeros) through the per-chunk scan: every chunk's stream == the oracle's stream of that chunk"""
    monkeypatch.setenv("X3H_SEG_MIN", "2")
    parts = [synth.english_like(900, seed=2).tobytes(), b"", b"x", bytes(700), synth.zipf_bytes(800, offset=99).tobytes()]
    data = np.frombuffer(b"".join(parts), np.uint8)
    off = np.cumsum([0] + [len(p) for p in parts]).astype(np.uint64)
    kw = dict(w_kib=1, t=3)
    got = emu.compress_chunks(data, off, _lib.make_params(**kw))
    for g, p in zip(got, parts):
        assert g == oracle.compress(p, oracle_lib.params(**kw))


def test_emulated_scan_counts(emu, oracle):
    data = synth.zipf_bytes(300).tobytes()
    cnt = emu.scan_counts(data, _lib.make_params(w_kib=1, t=2))
    for p in (0, 1, 150, 299):
        assert np.array_equal(cnt[p], oracle.count(data, p, 1024))


@pytest.mark.parametrize("name", ["empty", "one_byte", "abra_x100", "zeros1024", "quad4096", "records_w8_t16", "gpl8k_x", "rand4096_w4_t8"])
def test_emulated_decoder_reproduces_golden_inputs(emu, golden, name):
    """decode.hip (x3.c:285-353) on the reference's own streams."""
    c = golden[name]
    assert emu.decompress(c["expect"], len(c["data"]) + 8) == c["data"]


def test_emulated_decoder_random_streams(emu, oracle):
    """decode.hip on the ORACLE's streams of random inputs / parameters, singly and as ragged batches: dictionaries beyond the (emulator-sized)
    LDS tables, tags that follow themselves, pairs that repeat themselves, lists of more than 64 items, every capacity from exact to +8"""
    rng = np.random.default_rng(20260)
    text, zipf = synth.english_like(1 << 15, seed=3).tobytes(), synth.zipf_bytes(1 << 15, offset=77).tobytes()

    def gen(n):
        kind = int(rng.integers(0, 8))
        if n == 0: return b""
        o = int(rng.integers(0, (1 << 15) - n))
        if kind == 0: return text[o:o + n]
        if kind == 1: return zipf[o:o + n]
        if kind == 2: return bytes(rng.integers(0, 256, n, dtype=np.uint8))
        if kind == 3: return bytes(rng.integers(0, int(rng.integers(1, 5)), n, dtype=np.uint8))
        if kind == 4: q = bytes(rng.integers(0, 256, int(rng.integers(1, 40)), dtype=np.uint8)); return (q * (n // len(q) + 1))[:n]
        if kind == 5: return text[o:o + n // 2] + bytes(n - n // 2)
        if kind == 6: return synth.mr_like(n, seed=int(rng.integers(0, 1 << 30))).tobytes()
        a = np.zeros(n, np.uint8); k = max(1, n // 12); a[rng.integers(0, n, k)] = rng.integers(1, 4, k); return a.tobytes()

    for case in range(24):
        nch = int(rng.choice([1, 1, 2, 5]))
        sizes = [int(rng.choice([0, 1, 2, 33, 200, 700, 1500, 3000, 5000])) for _ in range(nch)]
        kw = dict(w_kib=int(rng.choice([1, 1, 2, 4])), t=int(rng.choice([0, 1, 2, 3, 8, 16, 64])), m=int(rng.choice([0, 1, 4, 4, 9])),
                  n=int(rng.choice([0, 0, 1, 2])), x=int(rng.random() < 0.15))
        parts = [gen(n) for n in sizes]
        streams = [oracle.compress(q, oracle_lib.params(**kw)) for q in parts]
        back = emu.decompress_chunks(streams, [n + int(rng.integers(0, 9)) for n in sizes]) if nch > 1 else [emu.decompress(streams[0], sizes[0] + 3)]
        assert back == parts, f"case {case}: sizes {sizes} params {kw}"


def test_emulated_decoder_reaches_its_rare_paths(emu, oracle):
    """decode.hip counts, in emulator builds, how often its rare paths run (x3emu_dec_cover[]): a context0 block that moved reached through the item that still holds
    its old place (forwarding entry followed, the item patched -- also in the lanes when its list is the current context1), symbols and tags beyond entry 63 of a
    list, blocks moving, ranks in the second register of the recency list (64-127: this build keeps two, the product four) and behind the registers.  Each input below is built for one of them; all of them must have run, and every round trip is exact."""
    import ctypes as C
    cover = (C.c_uint * 8).in_dll(emu.lib, "x3emu_dec_cover")
    for i in range(8):
        cover[i] = 0
    rng = np.random.default_rng(41)
    words = [bytes(rng.integers(97, 123, 3, dtype=np.uint8)) for _ in range(150)]            # 150 distinct 3-byte words: lists and ranks beyond 64
    vocab = b"".join(words[int(i)] for i in rng.integers(0, 150, 3000))
    zeros = bytes(400) + b"ab" + bytes(300) + b"cd" + bytes(500) + b"ab" + bytes(200)       # runs of one tag: pair (0, 0), self-contexts, fragments in between
    mixed = b"".join(bytes([int(a)]) * int(n) + words[int(w)] for a, n, w in zip(rng.integers(0, 3, 300), rng.integers(1, 9, 300), rng.integers(0, 20, 300)))
    for data, kw in ((vocab, dict(w_kib=8, t=1)), (zeros, dict(w_kib=1, t=2)), (mixed, dict(w_kib=2, t=2)), (synth.zipf_bytes(6000, offset=9).tobytes(), dict(w_kib=4, t=3))):
        assert emu.decompress(oracle.compress(data, oracle_lib.params(**kw)), len(data) + 5) == data
    got = [int(cover[i]) for i in range(7)]
    assert all(g > 0 for g in got), (f"rare paths not reached: forward {got[0]}, far decode {got[1]}, far find {got[2]}, block moved {got[3]}, rank behind the registers {got[4]}, "
                                     f"lane patch {got[5]}, rank in a register behind the first {got[6]}")


def test_emulated_device_resident_decode(emu, oracle):
    """x3h_decompress_chunks_dev: streams read and bytes written in place (the emulator's device memory is host memory)"""
    parts = [synth.english_like(900, seed=2).tobytes(), b"", synth.zipf_bytes(700, offset=5).tobytes(), b"z" * 300]
    kw = dict(w_kib=1, t=3)
    streams = [oracle.compress(q, oracle_lib.params(**kw)) for q in parts]
    blob = np.frombuffer(b"".join(streams), dtype=np.uint8).copy()
    ioff = np.cumsum([0] + [len(x) for x in streams]).astype(np.uint64)
    caps = [len(q) + 3 for q in parts]
    ooff = np.cumsum([0] + caps).astype(np.uint64)
    out = np.full(int(ooff[-1]), 0xEE, dtype=np.uint8)
    lens, _ = emu.decompress_chunks_dev(blob.ctypes.data, ioff, out.ctypes.data, ooff)
    assert [int(x) for x in lens] == [len(q) for q in parts]
    for i, q in enumerate(parts):
        assert out[int(ooff[i]):int(ooff[i]) + len(q)].tobytes() == q
        assert (out[int(ooff[i]) + len(q):int(ooff[i + 1])] == 0xEE).all()       # the slack of every capacity is untouched


def test_emulated_decoder_survives_damaged_streams(emu, oracle):
    """bit flips, overwritten bytes, truncation and plain noise: the decoder ends with a status (corrupt / output full) or some output --
    never a fault or a hang (the reference abort()s or overruns its buffer, ac.c:178, x3.c:621)"""
    rng = np.random.default_rng(99)
    base = [oracle.compress(synth.english_like(2500, seed=s).tobytes(), oracle_lib.params(w_kib=1, t=2)) for s in range(2)]
    base.append(oracle.compress(synth.zipf_bytes(2000, offset=7).tobytes(), oracle_lib.params(w_kib=1, t=1)))
    seen = set()
    for case in range(60):
        kind = case % 4
        if kind == 0:
            s = bytes(rng.integers(0, 256, int(rng.integers(0, 300)) * 4, dtype=np.uint8))
        else:
            b = bytearray(base[int(rng.integers(0, len(base)))])
            for _ in range(int(rng.integers(1, 5))):
                i = int(rng.integers(0, len(b)))
                if kind == 1: b[i] ^= 1 << int(rng.integers(0, 8))
                elif kind == 2: b[i] = int(rng.integers(0, 256))
                else: b = b[:max(4, (i // 4) * 4)]
            s = bytes(b)
        try:
            emu.decompress(s, int(rng.choice([0, 10, 3000, 50000])))
            seen.add(0)
        except _lib.X3Error as e:
            assert e.status in (-3, -4), e.status
            seen.add(e.status)
    assert -4 in seen and -3 in seen


def test_emulated_decoder_errors(emu, golden):
    with pytest.raises(_lib.X3Error) as e:
        emu.decompress(golden["zeros5000"]["expect"], 100)   # ratio > 64:1 -- the reference overruns its buffer here (x3.c:621)
    assert e.value.status == -3
    with pytest.raises(_lib.X3Error) as e:
        emu.decompress(b"\x12\x34\x56\x78" * 50, 1000)
    assert e.value.status == -4


# ---- alternative schedules of the same computation: must not change a bit -------------------------------------------------------
@pytest.fixture()
def emu_env(monkeypatch):
    """an emulator context created under the given environment (the library reads its switches at context creation / call time)"""
    subprocess.run(["make", "-C", os.path.join(HERE, "emu")], check=True, capture_output=True)
    made = []

    def make(**env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        ctx = _lib.X3Context(0, library=EMU_SO)
        made.append(ctx)
        return ctx
    yield make
    for c in made:
        c.close()


PIPE_CASES = [
    ("tiny", b"ab", dict()),
    ("abra", b"abracadabra" * 30, dict(w_kib=1, t=2)),
    ("english", synth.english_like(9000).tobytes(), dict(w_kib=2, t=8)),
    ("zipf", synth.zipf_bytes(4000).tobytes(), dict(w_kib=1, t=2)),
    ("zeros", bytes(3000), dict(w_kib=1, t=15)),
]


@pytest.mark.parametrize("name,data,kw", PIPE_CASES, ids=[c[0] for c in PIPE_CASES])
def test_emulated_pipelined_single_stream(emu_env, oracle, name, data, kw):
    """api.hip run_pipelined: the coding stage of growing prefixes (parse checkpoints at 2/8/26/62 %) + segment-wise coder."""
    ctx = emu_env(X3H_PIPE_MIN="1")
    assert ctx.compress(data, _lib.make_params(**kw)) == oracle.compress(data, oracle_lib.params(**kw))


@pytest.mark.parametrize("name,data,kw", PIPE_CASES[1:], ids=[c[0] for c in PIPE_CASES[1:]])
def test_emulated_pipelined_emit_behind_every_segment(emu_env, oracle, name, data, kw):
    """X3H_SEG_EMIT=1: each coder segment's bits written behind its recurrence by one workgroup per stream (the default for 8 or more streams)
    against the other form of the schedule's tail (operands and chain states gathered into the final layout, every bit emitted at the end)"""
    ctx = emu_env(X3H_PIPE_MIN="1", X3H_SEG_EMIT="1")
    assert ctx.compress(data, _lib.make_params(**kw)) == oracle.compress(data, oracle_lib.params(**kw))


@pytest.mark.parametrize("marks", ["0.5", "0.1,0.2,0.3,0.4,0.5,0.6,0.7,0.8", "0.9,0.95", "garbage"])
def test_emulated_pipelined_other_checkpoints(emu_env, oracle, marks):
    """X3H_PIPE_MARKS: one checkpoint, the maximum of eight, late ones, and a malformed list (keeps the default): same bytes"""
    ctx = emu_env(X3H_PIPE_MIN="1", X3H_PIPE_MARKS=marks)
    data, kw = synth.english_like(7000, seed=9).tobytes(), dict(w_kib=2, t=4)
    assert ctx.compress(data, _lib.make_params(**kw)) == oracle.compress(data, oracle_lib.params(**kw))


@pytest.mark.parametrize("pipe", ["0", "1"])
def test_emulated_fixed_point_modes(emu_env, oracle, pipe):
    """code2.hip modes_fixed_point (forced): the mode sequence as the fixed point of a chip-wide iteration."""
    ctx = emu_env(X3H_MODES="fixed", X3H_PIPE_MIN=pipe)
    data, kw = synth.english_like(20000).tobytes(), dict(w_kib=4, t=8)
    assert ctx.compress(data, _lib.make_params(**kw)) == oracle.compress(data, oracle_lib.params(**kw))
    assert ctx.last_stats.mode_iters > 0


@pytest.mark.parametrize("seg_emit", ["0", "1"], ids=["bits-at-the-end", "bits-behind-every-segment"])
def test_emulated_pipelined_several_streams(emu_env, oracle, seg_emit):
    """run_pipelined with a batch: ragged streams (an empty and a 2-byte one among them), per-stream checkpoints, per-stream coder segments
    in the operand / state rings, gather into the final symbol layout -- or, the other form of the tail, every segment's bits written behind it"""
    ctx = emu_env(X3H_PIPE_MIN="1", X3H_SEG_EMIT=seg_emit)
    kw = dict(w_kib=2, t=8)
    parts = [synth.english_like(9000).tobytes(), b"", synth.zipf_bytes(3000).tobytes(), b"ab", synth.english_like(20000, seed=5).tobytes(), bytes(2500)]
    data = np.frombuffer(b"".join(parts), dtype=np.uint8)
    off = np.cumsum([0] + [len(p) for p in parts]).astype(np.uint64)
    streams = ctx.compress_chunks(data, off, _lib.make_params(**kw))
    assert ctx.last_stats.pipelined == 1
    for p, got in zip(parts, streams):
        assert got == oracle.compress(p, oracle_lib.params(**kw))


def test_emulated_csb_count_kernel_body(emu_env, oracle):
    """the radix-4 count kernel itself (1024-thread workgroups) on the emulator, once; the other tests use its loop equivalent"""
    ctx = emu_env(X3_EMU_CSB_KERNEL="1")
    data, kw = synth.english_like(6000).tobytes(), dict(w_kib=2, t=8)
    assert ctx.compress(data, _lib.make_params(**kw)) == oracle.compress(data, oracle_lib.params(**kw))


def test_emulated_slices_arranged_by_the_per_stream_sort(emu_env, oracle):
    """K3 in slices with the two arrangements of a slice made by x3_segsort_kernel (X3H_SLICE_SEGSORT=1: one workgroup per stream, any segment length) instead of the
    LDS counting sort of small slices / the chip-wide sort of large ones"""
    ctx = emu_env(X3H_SLICED_MIN="1024", X3H_SLICE_GAP="700", X3H_SLICE_ARRANGE="0", X3H_SLICE_SEGSORT="1")
    data, kw = synth.english_like(5200, seed=8).tobytes(), dict(w_kib=2, t=6)
    assert ctx.compress(data, _lib.make_params(**kw)) == oracle.compress(data, oracle_lib.params(**kw))
    assert ctx.last_stats.pipelined == 2


@pytest.mark.parametrize("sched", ["stage-after-stage"])
def test_emulated_sort_and_scan_kernels(emu_env, oracle, sched):
    """prims.hip's hand-written radix sort (histogram / scan / ranked scatter) and prefix scans themselves on the emulator (X3_EMU_PRIM_KERNELS=1; the other tests
    answer these calls with host loops, for speed): K1 of a single stream sorts every padded position (several tiles of 4096, four 8-bit passes, the last tile
    ragged), the generic coding stage sorts and scans hits, touch events and symbols (odd bit counts: a narrower last pass)"""
    env = dict(X3_EMU_PRIM_KERNELS="1")
    env.update({"X3H_PIPE_MIN": "0"} if sched == "stage-after-stage" else {"X3H_SLICED_MIN": "1024", "X3H_SLICE_GAP": "700", "X3H_SLICE_ARRANGE": "0"})
    ctx = emu_env(**env)
    sets = [(synth.english_like(5200, seed=3).tobytes(), dict(w_kib=4, t=6))] if sched == "stage-after-stage" else [(synth.mr_like(3000).tobytes(), dict(w_kib=1, t=3))]
    for data, kw in sets:
        assert ctx.compress(data, _lib.make_params(**kw)) == oracle.compress(data, oracle_lib.params(**kw))


# ---- chunking behind the C boundary: sub-batches, several handles, X3C1 container (api.hip, x3_container.c) -------------------------
def test_emulated_sub_batching_by_padded_bytes(emu_env, oracle):
    """many small chunks under a large window: the sub-batches are cut on the PADDED layout (len + W + slack per chunk), which K1
    indexes with 32 bits -- not on the input bytes"""
    kw = dict(w_kib=16, t=8)
    parts = [synth.english_like(700 + 13 * i, seed=40 + i).tobytes() for i in range(7)] + [b""]
    data = np.frombuffer(b"".join(parts), dtype=np.uint8)
    off = np.cumsum([0] + [len(p) for p in parts]).astype(np.uint64)
    want = [oracle.compress(p, oracle_lib.params(**kw)) for p in parts]
    ctx = emu_env(X3H_BATCH_PAD_BYTES=str(3 * (16384 + 4096 + 1024)))   # room for two or three padded chunks per sub-batch
    assert ctx.compress_chunks(data, off, _lib.make_params(**kw)) == want
    assert ctx.last_stats.steps == sum(len(oracle.trace(p, oracle_lib.params(**kw))[1]) for p in parts)


def test_emulated_output_full_in_one_sub_batch_reports_every_length(emu_env):
    """a sub-batch that does not fit its capacity must not hide the others' results (every out_lens entry is written)"""
    import ctypes as C
    ctx = emu_env(X3H_BATCH_BYTES="3000")
    rnd = np.random.default_rng(3).integers(0, 256, 2500, dtype=np.uint8).tobytes()   # incompressible: needs > 2500 bytes
    parts = [bytes(2000), rnd, bytes(1500)]
    data = np.frombuffer(b"".join(parts), dtype=np.uint8)
    off = np.cumsum([0] + [len(p) for p in parts]).astype(np.uint64)
    stride, lens, st = 1024, np.full(3, 77, dtype=np.uint64), _lib.Stats()
    out = np.zeros(stride * 3, dtype=np.uint8)
    prm = _lib.make_params(w_kib=1, t=4)
    rc = ctx.lib.x3h_compress_chunks(ctx._h, C.byref(prm), data.ctypes.data, off.ctypes.data, 3, out.ctypes.data, stride, lens.ctypes.data, C.byref(st))
    assert rc == -3 and lens[0] > 0 and lens[2] > 0 and lens[0] < 100 and lens[2] < 100
    assert st.steps > 0


def test_emulated_container_round_trip_two_handles(emu_env, oracle):
    """x3h_compress_container / x3h_decompress_container over two handles: chunk c == the oracle's stream of chunk c, one chunk stays raw"""
    a, b = emu_env(X3H_MULTI_SERIAL="1"), emu_env()
    kw = dict(w_kib=2, t=8)
    prm, oprm = _lib.make_params(**kw), oracle_lib.params(**kw)
    data = synth.english_like(7001).tobytes()
    blob = _lib.compress_container([a, b], data, prm, 2048)
    params, chunks = __import__("x3_compressor_amd.container", fromlist=["x"]).unpack(blob)
    assert params["window_bytes"] == 2048 and params["max_match_count"] == 8 and len(chunks) == 4
    for i, (raw, s) in enumerate(chunks):
        assert raw == len(data[i * 2048:(i + 1) * 2048]) and s == oracle.compress(data[i * 2048:(i + 1) * 2048], oprm)
    assert _lib.decompress_container([a, b], blob, len(data)) == data
    with pytest.raises(_lib.X3Error) as e:
        _lib.decompress_container([a], blob, len(data) - 1)
    assert e.value.status == -3
    bad = bytearray(blob); bad[-5] ^= 0x40
    try:
        assert _lib.decompress_container([a], bytes(bad), len(data)) != data
    except _lib.X3Error as e2:
        assert e2.status == -4
    # one chunk: no frame at all, the reference's raw stream (x3.c:603-611)
    one = _lib.compress_container([a, b], data[:1500], prm, 2048)
    assert one == oracle.compress(data[:1500], oprm)
    assert _lib.decompress_container([a], one, 4096) == data[:1500]
    assert _lib.compress_container([a], b"", prm, 2048) == oracle.compress(b"", oprm)
    # chunks over handles directly
    off = np.array([0, 100, 100, 3000, 7001], dtype=np.uint64)
    got = _lib.compress_chunks_multi([a, b, a], np.frombuffer(data, np.uint8), off, prm)
    assert got == [oracle.compress(data[int(off[i]):int(off[i + 1])], oprm) for i in range(4)]


def test_emulated_coder_chain_seam(emu, oracle):
    """x3h_coder_chain (stage seam) on the emulator: the functional restatement of the chain == the oracle's E1/E2/E3 loops (the asm chain
    itself is checked by tests/test_gpu_parity.py::test_coder_chain_states_equal_oracle_intervals on the GPU)"""
    rng = np.random.default_rng(11)
    n = 777
    total = rng.integers(2, 2 ** 22, n)
    freq = np.minimum(rng.integers(1, total + 1), total)
    cum = np.minimum((rng.random(n) * (total - freq + 1)).astype(np.int64), total - freq)
    states, fin = emu.coder_chain(cum, freq, total)
    lo, hi = oracle.ac_chain(cum, freq, total)
    g = np.arange(1, (n + 7) // 8)
    assert np.array_equal(states[1:, 0], lo[8 * g - 1])
    assert np.array_equal(states[1:, 1].astype(np.int64), hi[8 * g - 1].astype(np.int64) - lo[8 * g - 1].astype(np.int64) + 1)
    assert fin == int(lo[-1])


def test_emulated_pipelined_resumed_mode_chain(emu_env, oracle):
    """growing prefixes with the serial mode kernel RESUMING behind the hits of the previous prefix (its saved chain state + the earlier
    modes re-laid out for the new per-stream counts): streams whose dictionaries fit the kernel's LDS table (48 entries in this build)"""
    ctx = emu_env(X3H_PIPE_MIN="1", X3H_MODES="serial")
    kw = dict(w_kib=1, t=3)
    parts = [b"abracadabra" * 300, bytes(3000), (b"abcabdabe" * 400)[:3500], b"xyzzy" * 500, b"ab"]
    data = np.frombuffer(b"".join(parts), dtype=np.uint8)
    off = np.cumsum([0] + [len(p) for p in parts]).astype(np.uint64)
    streams = ctx.compress_chunks(data, off, _lib.make_params(**kw))
    assert ctx.last_stats.pipelined == 1 and ctx.last_stats.dict_elems < 48 * 5
    for p, got in zip(parts, streams):
        assert got == oracle.compress(p, oracle_lib.params(**kw))


def _dense_cases():
    rng = np.random.default_rng(3)
    sparse = np.zeros(3000, np.uint8)
    sparse[rng.integers(0, 3000, 250)] = rng.integers(1, 4, 250)
    return [
        ("zipf_endzeros", synth.zipf_bytes(1500).tobytes() + bytes(40), dict(w_kib=1, t=3)),
        ("zeros_t2", bytes(700), dict(w_kib=1, t=2)),
        ("sparse", sparse.tobytes(), dict(w_kib=1, t=4)),
        ("mr", synth.mr_like(4000).tobytes(), dict(w_kib=2, t=6)),
        ("text_then_zeros", synth.english_like(1200).tobytes() + bytes(300), dict(w_kib=1, t=3)),
        ("zero_then_text", bytes(200) + synth.english_like(1200).tobytes() + b"\0", dict(w_kib=1, t=3)),
        ("periodic", b"\0\0\0\1" * 600, dict(w_kib=1, t=5)),
    ]


@pytest.mark.parametrize("name,data,kw", _dense_cases(), ids=[c[0] for c in _dense_cases()])
def test_emulated_scan_dense_classes_and_padding(emu, oracle, name, data, kw):
    """K1 with DENSE classes (this build: more than 6 members of a class inside a window): byte-by-byte refinement instead of the sweep,
    padding positions counted analytically once they are dropped from the lists -- m[] and the stream against the oracle"""
    prm, oprm = _lib.make_params(**kw), oracle_lib.params(**kw)
    assert np.array_equal(emu.scan_m(data, prm), oracle.scan_m(data, oprm))
    assert emu.compress(data, prm) == oracle.compress(data, oprm)


# ---- per-stream kernels of code3.hip (tokens, move-to-front ranks, context statistics, mode state, order-0 models, bit emission) -----------
@pytest.mark.parametrize("name,data,kw", CASES + PIPE_CASES[2:4], ids=[c[0] for c in CASES] + ["english9k", "zipf4k"])
def test_emulated_stream_kernels_forced(emu_env, oracle, name, data, kw):
    """X3H_STREAM_KERNELS=1: the per-stream LDS kernels on single streams (they are the default only for batches of >= 48 streams)"""
    ctx = emu_env(X3H_STREAM_KERNELS="1", X3H_PIPE_MIN="0", X3H_ARRANGE="1" if len(data) % 2 else "0", X3H_SEGSORT="1" if len(data) % 3 else "0")
    assert ctx.compress(data, _lib.make_params(**kw)) == oracle.compress(data, oracle_lib.params(**kw))


def test_emulated_stream_kernels_batch_of_52(emu_env, oracle):
    """a ragged batch of 52 streams takes the many-stream schedule by default: token walk kernel, side-stream move-to-front ranks, context
    kernel with several wavefronts per stream (cut at context boundaries), emit kernel with carried pending bits"""
    rng = np.random.default_rng(5)
    parts = []
    for i in range(52):
        n, kind = int(rng.integers(0, 220)), i % 5
        if kind == 0: parts.append(synth.english_like(n + 200, seed=i).tobytes())
        elif kind == 1: parts.append(synth.zipf_bytes(n, offset=100 * i).tobytes())
        elif kind == 2: parts.append(bytes(n))
        elif kind == 3: parts.append(rng.integers(0, 256, n, dtype=np.uint8).tobytes())
        else: parts.append((b"abcab" * 200)[:n])
    parts[7], parts[20] = b"", b"x"
    kw = dict(w_kib=1, t=3)
    data = np.frombuffer(b"".join(parts), dtype=np.uint8)
    off = np.cumsum([0] + [len(p) for p in parts]).astype(np.uint64)
    # (K1 by one emulated 1024-thread workgroup per chunk costs ~1.5 s a chunk: it has its own tests above, here the chip-wide scan runs)
    want = [oracle.compress(p, oracle_lib.params(**kw)) for p in parts]
    streams = emu_env(X3H_SEG_MIN="1000").compress_chunks(data, off, _lib.make_params(**kw))
    for i, got in enumerate(streams):
        assert got == want[i], f"stream {i}"
    # a fixed number of wavefronts per stream in the context kernel, no XCD mapping -- on the first sixteen streams (emulator time)
    sub = emu_env(X3H_SEG_MIN="1000", X3H_CTX_SUB="3", X3H_CTX_XCD="0", X3H_STREAM_KERNELS="1", X3H_PIPE_MIN="0").compress_chunks(data[:int(off[16])], off[:17], _lib.make_params(**kw))
    for i in range(16):
        assert sub[i] == want[i], f"X3H_CTX_SUB=3: stream {i}"
    # X3H_ARRANGE=1: hits arranged by one workgroup per stream (x3_arrange_kernel) instead of the chip-wide sort -- on the first ten streams (emulator time)
    few = emu_env(X3H_ARRANGE="1", X3H_STREAM_KERNELS="1", X3H_PIPE_MIN="0").compress_chunks(data[:int(off[10])], off[:11], _lib.make_params(**kw))
    for i in range(10):
        assert few[i] == want[i], f"per-stream arrangement: stream {i}"
    # X3H_SEGSORT=1: the arrangements by one workgroup per stream in 8-bit passes (x3_segsort_kernel; the default from 128 streams on) -- the first twelve streams
    few = emu_env(X3H_SEGSORT="1", X3H_STREAM_KERNELS="1", X3H_PIPE_MIN="0").compress_chunks(data[:int(off[12])], off[:13], _lib.make_params(**kw))
    for i in range(12):
        assert few[i] == want[i], f"per-stream sort: stream {i}"
    few = emu_env(X3H_SEGSORT="1", X3H_SEGSORT_NINE="1", X3H_STREAM_KERNELS="1", X3H_PIPE_MIN="0").compress_chunks(data[:int(off[12])], off[:13], _lib.make_params(**kw))
    for i in range(12):
        assert few[i] == want[i], f"per-stream sort, one pass of nine bits: stream {i}"
    few = emu_env(X3H_SEGSORT="1", X3H_SEGSORT_NINE="0", X3H_SEGSORT_PASSES="3", X3H_SEGSORT_GEN="1", X3H_STREAM_KERNELS="1", X3H_PIPE_MIN="0").compress_chunks(data[:int(off[12])], off[:13], _lib.make_params(**kw))
    for i in range(12):
        assert few[i] == want[i], f"per-stream sort: stream {i}"


@pytest.mark.parametrize("env", [{}, dict(X3H_PIPE_MIN="1"), dict(X3H_PIPE_MIN="1", X3H_SEG_EMIT="1"), dict(X3H_STREAM_KERNELS="1")],
                         ids=["stage-after-stage", "pipelined", "pipelined-segment-emit", "stream-kernels"])
def test_emulated_size_estimates_match_oracle(oracle, monkeypatch, env):
    """x3h_stats.est_bits (x3h_ctx_set_estimates; x3.c:43,192-193,253-266): the terms made by the symbol assembly pass and summed per stream in coding
    order by x3_est_kernel, under every schedule -- equal to the oracle's accumulators, which equal the real reference's statistics lines
    (tests/test_oracle_golden.py).  On the CPU the term -log2(double) -> float and glibc's log2f may differ by an ulp: 1e-6 relative."""
    subprocess.run(["make", "-C", os.path.join(HERE, "emu")], check=True, capture_output=True)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    with _lib.X3Context(0, library=EMU_SO) as ctx:
        ctx.set_estimates(True)
        for data, kw in ((synth.english_like(2600).tobytes(), dict(w_kib=1, t=4)), (synth.zipf_bytes(1500).tobytes(), dict(w_kib=1, t=2)),
                         (bytes(1200), dict(w_kib=1, t=15)), (b"", dict()), (b"a", dict())):
            want, ost = oracle.compress(data, oracle_lib.params(**kw), want_stats=True)
            assert ctx.compress(data, _lib.make_params(**kw)) == want
            got = list(ctx.last_stats.est_bits)
            assert all(abs(g - w) <= 1e-6 * max(abs(w), 1.0) for g, w in zip(got, ost.sizes)), (got, list(ost.sizes))
        parts = [synth.english_like(900, seed=2).tobytes(), b"", b"x", synth.zipf_bytes(800, offset=99).tobytes()]
        off = np.cumsum([0] + [len(p) for p in parts]).astype(np.uint64)
        ctx.compress_chunks(np.frombuffer(b"".join(parts), np.uint8), off, _lib.make_params(w_kib=1, t=3))
        got = list(ctx.last_stats.est_bits)
        want = [0.0] * 4
        for p in parts:
            _, ost = oracle.compress(p, oracle_lib.params(w_kib=1, t=3), want_stats=True)
            want = [a + float(b) for a, b in zip(want, ost.sizes)]
        assert all(abs(g - w) <= 1e-6 * max(abs(w), 1.0) for g, w in zip(got, want)), (got, want)


# ---- K3 in slices (code4.hip; api.hip run_sliced) ---------------------------------------------------------------------------------
ALL_MARKS = "0.02,0.05,0.10,0.17,0.26,0.36,0.47,0.59,0.72,0.86"
SLICED_CASES = [
    ("abra_two_slices", b"abracadabra" * 30, dict(w_kib=1, t=2), "100", None),            # the first slice only inserts elements (no hit): recency order and index model still move
    ("english_eleven_slices", synth.english_like(6000, seed=8).tobytes(), dict(w_kib=2, t=1), "200", "3"),
    ("zipf_many_pairs", synth.zipf_bytes(5000).tobytes(), dict(w_kib=1, t=2), "150", "2"),
    ("zeros", bytes(3000), dict(w_kib=1, t=15), "100", None),
    ("mr_like", synth.mr_like(3000).tobytes(), dict(w_kib=1, t=3), "100", "7"),
    ("one_byte", b"a", dict(), "100", None),
]


@pytest.mark.parametrize("name,data,kw,gap,sub", SLICED_CASES, ids=[c[0] for c in SLICED_CASES])
def test_emulated_sliced_schedule_matches_oracle(oracle, monkeypatch, name, data, kw, gap, sub):
    """K3 in slices forced on small inputs: the parse's checkpoints cut a stream into up to eleven slices (X3H_SLICE_GAP = smallest distance of two marks),
    each slice goes through the per-stream feature kernels with the state of the earlier slices carried (last touches, context item lists in their pools,
    pair ordinals, model counters, order-0 models, coder interval, pending bits) -- stream, statistics and size estimates == the oracle's"""
    subprocess.run(["make", "-C", os.path.join(HERE, "emu")], check=True, capture_output=True)
    monkeypatch.setenv("X3H_SLICED_MIN", "1")
    monkeypatch.setenv("X3H_SLICE_MARKS", ALL_MARKS)   # (a short stream gets one mark by default: a slice has a fixed cost)
    monkeypatch.setenv("X3H_SLICE_GAP", gap)
    if sub:
        monkeypatch.setenv("X3H_SLICE_SUB", sub)
    with _lib.X3Context(0, library=EMU_SO) as ctx:
        ctx.set_estimates(True)
        want, ost = oracle.compress(data, oracle_lib.params(**kw), want_stats=True)
        assert ctx.compress(data, _lib.make_params(**kw)) == want
        st = ctx.last_stats
        assert st.pipelined == 2
        assert list(st.events)[:4] == list(ost.events)[:4] and st.ctx0_entries == ost.ctx0_entries and st.dict_elems == ost.dict_elems and st.steps == ost.steps
        assert all(abs(g - w) <= 1e-6 * max(abs(w), 1.0) for g, w in zip(st.est_bits, ost.sizes))


def test_emulated_sliced_ragged_batch(oracle, monkeypatch):
    """streams of different lengths end in different slices (an ended stream keeps its E_EOF symbol and codes nothing more), an empty and a one-byte
    stream in between; every stream == the oracle's stream of that chunk alone"""
    subprocess.run(["make", "-C", os.path.join(HERE, "emu")], check=True, capture_output=True)
    monkeypatch.setenv("X3H_SLICED_MIN", "1")
    monkeypatch.setenv("X3H_SLICE_MARKS", ALL_MARKS)
    monkeypatch.setenv("X3H_SLICE_GAP", "100")
    parts = [synth.english_like(4000, seed=2).tobytes(), b"", b"x", synth.zipf_bytes(2500, offset=99).tobytes(), synth.english_like(6000, seed=9).tobytes(), bytes(1500)]
    off = np.cumsum([0] + [len(p) for p in parts]).astype(np.uint64)
    kw = dict(w_kib=1, t=3)
    with _lib.X3Context(0, library=EMU_SO) as ctx:
        got = ctx.compress_chunks(np.frombuffer(b"".join(parts), np.uint8), off, _lib.make_params(**kw))
        assert ctx.last_stats.pipelined == 2
    for g, p in zip(got, parts):
        assert g == oracle.compress(p, oracle_lib.params(**kw))


def test_emulated_sliced_falls_back_on_a_large_dictionary(oracle, monkeypatch):
    """a dictionary beyond the sliced kernels' LDS tables (X3S_DMAX): run_sliced gives up after the parse and the batch is coded stage after stage"""
    subprocess.run(["make", "-C", os.path.join(HERE, "emu")], check=True, capture_output=True)
    monkeypatch.setenv("X3H_SLICED_MIN", "1")
    monkeypatch.setenv("X3H_SLICE_GAP", "500")
    # 2300 distinct three-byte records, each written three times in a row: at -t 1 every record becomes a dictionary element of its own
    recs = [bytes([i % 251, (i // 251) % 241 + 1, (i * 7) % 239 + 3]) for i in range(2300)]
    data = b"".join(r * 3 for r in recs)
    kw = dict(w_kib=1, t=1)
    with _lib.X3Context(0, library=EMU_SO) as ctx:
        want, ost = oracle.compress(data, oracle_lib.params(**kw), want_stats=True)
        assert ost.dict_elems > 2048
        assert ctx.compress(data, _lib.make_params(**kw)) == want
        assert ctx.last_stats.pipelined == 0 and ctx.last_stats.dict_elems == ost.dict_elems


@pytest.mark.parametrize("ndev", [2, 3, 8])
def test_emulated_rccl_container_over_several_devices(monkeypatch, oracle, ndev):
    """x3h_compress_container_rccl at ndevices = 2, 3, 8 (one rank: tests/test_gpu_parity.py, through the real RCCL; VERDICT r04: the C multi-rank code had only ever run as a one-rank self-send): N emulated devices
    (tests/emu/hip_shim.h, X3EMU_DEVICES), a handle each, librccl replaced by tests/emu/rccl_stub.h -- sends and receives recorded inside the group, paired and
    executed as memcpy at ncclGroupEnd, failing on any unmatched operation or size mismatch.  What runs is api.hip's own partition of the chunks over the devices, the
    per-device pack, the offsets `at` of every block behind the header, the send / receive pairing and the header placement; the bytes must equal
    x3h_compress_container's (host-staged concat), every chunk the oracle's stream, and the exchange must be ONE group of ndev - 1 pairs with exact sizes."""
    import ctypes as C
    subprocess.run(["make", "-C", os.path.join(HERE, "emu")], check=True, capture_output=True)
    monkeypatch.setenv("X3EMU_DEVICES", str(ndev))
    monkeypatch.setenv("X3H_MULTI_SERIAL", "1")   # the fiber emulator runs one device at a time on the calling thread
    ctxs = [_lib.X3Context(d, library=EMU_SO) for d in range(ndev)]
    try:
        lib = ctxs[0].lib
        kw = dict(w_kib=1, t=4)
        prm, oprm = _lib.make_params(**kw), oracle_lib.params(**kw)
        from x3_compressor_amd import container
        # 11 chunks (the last one short) over ndev devices: uneven blocks at 2 and 3 devices, devices with ONE chunk and a 3-chunk tail at 8; then fewer chunks than devices
        # (the emulator pays per workgroup it runs, so the chunk counts are small: five chunks -- 3 + 2 / 2 + 2 + 1 -- for two and three devices)
        sets = [(synth.english_like(4 * 600 + 250, seed=9).tobytes(), 600)] if ndev <= 3 else [(synth.english_like(10 * 500 + 250, seed=9).tobytes(), 500), (synth.zipf_bytes(3 * 400).tobytes(), 400)]
        for data, cb in sets:
            nch = (len(data) + cb - 1) // cb
            groups0, pairs, nbytes = C.c_uint64(), C.c_uint64(), C.c_uint64()
            lib.x3emu_rccl_last_group(C.byref(groups0), C.byref(pairs), C.byref(nbytes))
            got = _lib.compress_container(ctxs, data, prm, cb, rccl=True)
            groups1 = C.c_uint64()
            lib.x3emu_rccl_last_group(C.byref(groups1), C.byref(pairs), C.byref(nbytes))
            # what x3h_compress_container (host-staged concat) writes for these bytes, put together here: the X3C1 header + the oracle's stream of every chunk
            # (test_emulated_container_round_trip_two_handles checks that path itself against the same streams)
            streams = [oracle.compress(data[i * cb:(i + 1) * cb], oprm) for i in range(nch)]
            want = container.header([len(data[i * cb:(i + 1) * cb]) for i in range(nch)], [len(x) for x in streams], prm) + b"".join(streams)
            assert got == want
            params, chunks = container.unpack(got)
            assert len(chunks) == nch and [x for _, x in chunks] == streams
            nd = min(ndev, nch)
            assert groups1.value == groups0.value + 1, "the final concat is ONE send/receive group"
            assert pairs.value == (1 if nd == 1 else nd - 1)
            # exact sizes: what travels is every block but the root's (one rank: its own block, to itself)
            base, rem = divmod(nch, nd)   # api.hip shard() == dist.shard_range
            lo_hi = [(d * base + min(d, rem), d * base + min(d, rem) + base + (1 if d < rem else 0)) for d in range(nd)]
            sizes = [sum(len(chunks[i][1]) for i in range(lo, hi)) for lo, hi in lo_hi]
            assert sum(sizes) == len(got) - lib.x3h_container_header_bytes(nch)
            assert nbytes.value in (sum(sizes[1:]) if nd > 1 else sizes[0],), (nbytes.value, sizes)
            if ndev == 3:
                assert _lib.decompress_container(ctxs, got, len(data)) == data
    finally:
        ctxs[0].lib.x3h_rccl_release()
        for c in ctxs:
            c.close()
