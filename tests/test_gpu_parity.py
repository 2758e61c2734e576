"""Parity tests proper: the HIP library (through the C ABI) against the golden vectors of the real reference and
against the oracle on seeded inputs.  Bit-exact is the bar everywhere (byte / integer / index work)."""
import os

import numpy as np
import pytest
import torch  # before libx3hip.so is loaded: torch brings its own HIP runtime, and the process must end up with one (the first one loaded)

import golden_util
import oracle_lib
from x3_compressor_amd import _lib, synth

pytestmark = pytest.mark.gpu
CASES = sorted(golden_util.load_cases().keys())


@pytest.fixture(scope="module")
def gpu():
    ctx = _lib.X3Context(0)  # raises loudly if libx3hip.so or the GPU is missing
    yield ctx
    ctx.close()


def first_diff(a, b):
    a, b = np.asarray(a), np.asarray(b)
    n = min(len(a), len(b))
    d = np.nonzero(a[:n] != b[:n])[0]
    return int(d[0]) if len(d) else (n if len(a) != len(b) else -1)


# ---- whole path vs the reference's own outputs --------------------------------------------------------------
@pytest.mark.parametrize("name", CASES)
def test_stream_equals_reference_golden(gpu, golden, name):
    c = golden[name]
    got = gpu.compress(c["data"], _lib.params_from_args(c["args"]))
    assert got == c["expect"], f"first differing byte {first_diff(np.frombuffer(got, np.uint8), np.frombuffer(c['expect'], np.uint8))}"


# ---- stage level --------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["gpl16k_default", "quad4096_w2_t4", "zeros5000", "records_w8_t16", "cfg4_zipf32k_w64_t256",
                                  "cfg2_english32k_w64_t256", "rand4096_w4_t8", "gpl8k_t0", "range256_x8_w1_t1"])
def test_scan_m_equals_oracle(gpu, oracle, golden, name):
    c = golden[name]
    m = gpu.scan_m(c["data"], _lib.params_from_args(c["args"]))
    mo = oracle.scan_m(c["data"], oracle_lib.params_from_args(c["args"]))
    assert np.array_equal(m, mo), f"first diff at position {first_diff(m, mo)}"


@pytest.mark.parametrize("name", ["gpl16k_default", "records_w8_t16", "zeros1024"])
def test_scan_counts_equal_oracle(gpu, oracle, golden, name):
    """backend.c:62-74: the full 32-bin histogram, not just the selected length."""
    c = golden[name]
    prm = _lib.params_from_args(c["args"])
    cnt = gpu.scan_counts(c["data"], prm)
    rng = np.random.default_rng(3)
    for p in [0, 1, len(c["data"]) - 1, *rng.integers(0, len(c["data"]), 24)]:
        assert np.array_equal(cnt[p], oracle.count(c["data"], int(p), prm.window_bytes)), f"position {p}"


@pytest.mark.parametrize("name", [n for n in CASES if not n.startswith(("cfg3", "cfg5"))])
def test_parse_tokens_equal_oracle(gpu, oracle, golden, name):
    c = golden[name]
    tp, ti, d = gpu.parse(c["data"], _lib.params_from_args(c["args"]))
    _, op, oi, st = oracle.trace(c["data"], oracle_lib.params_from_args(c["args"]))
    assert d == st.dict_elems
    assert np.array_equal(tp, op), f"positions diverge at step {first_diff(tp, op)}"
    assert np.array_equal(ti, oi), f"tokens diverge at step {first_diff(ti, oi)}"


# ---- seeded inputs vs the oracle ------------------------------------------------------------------------------
SEEDED = [
    ("english96k_w64_t256", lambda: synth.english_like(96 * 1024).tobytes(), dict(w_kib=64, t=256)),
    ("zipf64k_w64_t256", lambda: synth.zipf_bytes(64 * 1024, offset=12345).tobytes(), dict(w_kib=64, t=256)),
    ("english24k_w256_t1024", lambda: synth.english_like(24 * 1024, seed=7).tobytes(), dict(w_kib=256, t=1024)),
    ("binary_mix_w16_t32", lambda: (np.random.default_rng(9).integers(0, 4, 50000, dtype=np.uint8) * 17).tobytes() + bytes(3000)
                                   + bytes(range(256)) * 20, dict(w_kib=16, t=32)),
    ("block_edge_2048", lambda: synth.english_like(2048).tobytes(), dict(w_kib=2, t=4)),
    ("block_edge_2049", lambda: synth.english_like(2049).tobytes(), dict(w_kib=2, t=4)),
    ("tiny_window_w0", lambda: synth.english_like(3000).tobytes(), dict(w_kib=1, t=1)),
    ("many_fragments_t1", lambda: np.random.default_rng(4).integers(0, 256, 40000, dtype=np.uint8).tobytes(), dict(w_kib=4, t=1)),
]


@pytest.mark.parametrize("name,make,kw", SEEDED, ids=[s[0] for s in SEEDED])
def test_stream_equals_oracle_on_seeded_input(gpu, oracle, name, make, kw):
    data = make()
    got = gpu.compress(data, _lib.make_params(**kw))
    want, st = oracle.compress(data, oracle_lib.params(**kw), want_stats=True)
    assert got == want, f"first differing byte {first_diff(np.frombuffer(got, np.uint8), np.frombuffer(want, np.uint8))}"
    gs = gpu.last_stats
    assert list(gs.events)[:4] == list(st.events)[:4] and gs.dict_elems == st.dict_elems
    assert gs.ctx0_entries == st.ctx0_entries and gs.steps == st.steps


@pytest.mark.parametrize("env", [{}, dict(X3H_PIPE_MIN="1"), dict(X3H_PIPE_MIN="1", X3H_SEG_EMIT="1"), dict(X3H_STREAM_KERNELS="1")],
                         ids=["stage-after-stage", "pipelined", "pipelined-segment-emit", "stream-kernels"])
def test_size_estimates_equal_oracle(gpu_env, oracle, env):
    """x3h_stats.est_bits (x3h_ctx_set_estimates): the reference's float accumulators sizes[] (x3.c:43,192-193,253-266) under every schedule.
    The oracle's are bit-identical to the real reference's (tests/test_oracle_golden.py); the GPU's terms are -log2 in double rounded to single, which
    differs from glibc's log2f by an ulp once in a while: 1e-6 relative."""
    ctx = gpu_env(**env)
    ctx.set_estimates(True)
    for data, kw in ((synth.english_like(200_000, seed=3).tobytes(), dict(w_kib=8, t=16)),
                     (synth.zipf_bytes(120_000, offset=7 << 20).tobytes(), dict(w_kib=64, t=256)),
                     (synth.mr_like(150_000, seed=5).tobytes(), dict(w_kib=8, t=15)), (b"", dict()), (b"a", dict())):
        want, ost = oracle.compress(data, oracle_lib.params(**kw), want_stats=True)
        assert ctx.compress(data, _lib.make_params(**kw)) == want
        got = list(ctx.last_stats.est_bits)
        for g, w in zip(got, ost.sizes):
            assert abs(g - w) <= 1e-6 * max(abs(w), 1.0), (got, list(ost.sizes))
    # a batch: the streams' accumulators added up
    text = synth.english_like(300_000, seed=11)
    off = np.array([0, 100_000, 100_000, 220_000, 300_000], dtype=np.uint64)
    prm = _lib.make_params(w_kib=8, t=16)
    ctx.compress_chunks(text, off, prm)
    got = list(ctx.last_stats.est_bits)
    want = [0.0] * 4
    for i in range(4):
        _, ost = oracle.compress(text[int(off[i]):int(off[i + 1])].tobytes(), oracle_lib.params(w_kib=8, t=16), want_stats=True)
        want = [a + float(b) for a, b in zip(want, ost.sizes)]
    assert all(abs(g - w) <= 1e-6 * max(abs(w), 1.0) for g, w in zip(got, want)), (got, want)
    ctx.set_estimates(False)
    ctx.compress(text[:50_000], prm)
    assert list(ctx.last_stats.est_bits) == [0.0] * 4


def test_chunks_are_independent_streams(gpu, oracle):
    """chunk output == `x3 -z` of that chunk alone; ragged sizes incl. an empty chunk."""
    data = synth.english_like(70000).tobytes()
    cuts = [0, 20000, 20000, 20001, 45000, 70000]
    kw = dict(w_kib=8, t=16)
    outs = gpu.compress_chunks(data, cuts, _lib.make_params(**kw))
    for i, o in enumerate(outs):
        assert o == oracle.compress(data[cuts[i]:cuts[i + 1]], oracle_lib.params(**kw)), f"chunk {i}"


def test_output_capacity_is_checked(gpu):
    data = np.random.default_rng(1).integers(0, 256, 5000, dtype=np.uint8).tobytes()
    with pytest.raises(_lib.X3Error) as e:
        gpu.compress(data, _lib.make_params(), cap=1000)
    assert e.value.status == -3  # X3H_E_OUTPUT_FULL; the reference would overrun its 2n buffer (x3.c:580)


# ---- BASELINE.json full size: size-independent properties ---------------------------------------------------------
def test_full_size_dickens_like_round_trip(gpu, oracle):
    """config 2 shape (10 192 446 bytes, -w 64 -t 256): the oracle's decoder (restating x3.c:285-353) must reproduce the
    input from the GPU stream, and the stream must be self-consistent (word padded, events add up)."""
    data = synth.english_like(synth.DICKENS_BYTES)
    stream = gpu.compress(data, _lib.make_params(w_kib=64, t=256))
    st = gpu.last_stats
    assert len(stream) % 4 == 0 and 0 < len(stream) < len(data)
    assert sum(list(st.events)[:4]) == st.steps
    rc, back = oracle.decompress(stream, len(data) + 64)
    assert rc == 0 and back == data.tobytes()
    assert gpu.decompress(stream, len(data)) == back  # and the GPU decoder agrees


# ---- decoder (x3.c:285-353, ac.c:128-198) ---------------------------------------------------------------------------
@pytest.mark.parametrize("name", CASES)
def test_decoder_reproduces_golden_inputs(gpu, golden, name):
    c = golden[name]
    assert gpu.decompress(c["expect"], len(c["data"]) + 16) == c["data"]


def test_decoder_capacity_and_corruption(gpu, golden):
    with pytest.raises(_lib.X3Error) as e:
        gpu.decompress(golden["zeros5000"]["expect"], 100)  # > 64:1: the reference overruns its output buffer (x3.c:621)
    assert e.value.status == -3
    with pytest.raises(_lib.X3Error) as e:
        gpu.decompress(bytes(range(7, 251)) * 4, 1 << 16)
    assert e.value.status == -4
    assert gpu.decompress(golden["empty"]["expect"], 0) == b""


def test_round_trip_config5_shape(gpu):
    """BASELINE config 5 shape at reduced size: 16-bit image-like data, -w 512 -t 4096, compress + decompress on the GPU."""
    rng = np.random.default_rng(55)
    x = np.cumsum(rng.integers(-6, 7, 400_000), dtype=np.int64)
    img = ((x - x.min()) % 4096).astype("<u2")  # smooth 12-bit samples in 16-bit words, like MR slices
    data = img.tobytes()
    stream = gpu.compress(data, _lib.make_params(w_kib=512, t=4096))
    assert gpu.decompress(stream, len(data)) == data


def test_chunked_round_trip(gpu):
    data = synth.zipf_bytes(300_000).tobytes()
    cuts = list(range(0, 300_000, 65536)) + [300_000]
    prm = _lib.make_params(w_kib=64, t=256)
    streams = gpu.compress_chunks(data, cuts, prm)
    back = gpu.decompress_chunks(streams, [cuts[i + 1] - cuts[i] for i in range(len(cuts) - 1)])
    assert b"".join(back) == data


# ---- K1 v2 (sorted n-gram lists) against the brute-force sweep kernel and the oracle --------------------------------------
def test_scan_v2_equals_brute_force_kernel(gpu, monkeypatch):
    monkeypatch.setenv("X3H_SCAN_V1", "1")
    brute = _lib.X3Context(0)  # reads the switch at creation
    monkeypatch.delenv("X3H_SCAN_V1")
    rng = np.random.default_rng(21)
    cases = [
        (synth.english_like(400_000).tobytes(), dict(w_kib=64, t=256)),
        (synth.zipf_bytes(300_000).tobytes(), dict(w_kib=64, t=256)),
        ((rng.integers(0, 4, 200_000, dtype=np.uint8) + 65).tobytes(), dict(w_kib=16, t=40)),          # DNA-like: deep walks
        ((b"abcdefghijklmnopqrstuvwxyz0123456789" * 6000)[:200_000], dict(w_kib=32, t=100)),             # periodic: lcp 32 everywhere
        (bytes(100_000) + synth.english_like(50_000).tobytes() + bytes(3000), dict(w_kib=8, t=15)),        # zero runs + text + zero tail
        (synth.english_like(120_000).tobytes(), dict(w_kib=512, t=4096)),                                   # window > input
        (synth.english_like(50_000).tobytes(), dict(w_kib=1, t=1)),
    ]
    try:
        for data, kw in cases:
            prm = _lib.make_params(**kw)
            a, b = gpu.scan_m(data, prm), brute.scan_m(data, prm)
            assert np.array_equal(a, b), f"{kw}: first diff at {first_diff(a, b)}"
    finally:
        brute.close()


def test_large_dictionary_60k_elements(gpu, oracle):
    """D = 60 265 > the 32 768 ranks of the LDS index-model table: the generic (global memory) path of the modes kernel, a hash
    table that doubled seven times in K2, and the decoder's large tables."""
    rng = np.random.default_rng(78)
    words = rng.integers(0, 256, (60000, 5), dtype=np.uint8)
    data = np.repeat(words, 4, axis=0).reshape(-1).tobytes()  # every 5-byte word four times in a row
    kw = dict(w_kib=1, t=1)
    want, st = oracle.compress(data, oracle_lib.params(**kw), want_stats=True)
    assert st.dict_elems > 32768
    got = gpu.compress(data, _lib.make_params(**kw))
    assert got == want, f"first differing byte {first_diff(np.frombuffer(got, np.uint8), np.frombuffer(want, np.uint8))}"
    assert gpu.last_stats.dict_elems == st.dict_elems
    assert gpu.decompress(got, len(data)) == data


# ---- BASELINE configs 3 and 5 at (near) full shape: size-independent properties ----------------------------------------------
SILESIA_SIZES = [10192446, 51220480, 9970564, 33553445, 6152192, 10085684, 6627202, 21606400, 7251944, 41458703, 8474240, 5345280]


def _mr_like(n, seed=55):
    rng = np.random.default_rng(seed)
    x = np.cumsum(rng.integers(-6, 7, n // 2 + 1), dtype=np.int64)
    return ((x - x.min()) % 4096).astype("<u2").tobytes()[:n]


def test_config3_shape_twelve_streams_w256_t1024(gpu):
    """12 independent streams with the Silesia size profile (1/16 scale, mixed content), -w 256 -t 1024, one batch:
    every stream decodes back to its input on the GPU, and a stream of the batch equals its standalone compression."""
    sizes = [s // 16 for s in SILESIA_SIZES]
    parts = []
    for i, n in enumerate(sizes):
        if i % 3 == 0: parts.append(synth.english_like(n, seed=100 + i).tobytes())
        elif i % 3 == 1: parts.append(synth.zipf_bytes(n, offset=7919 * i).tobytes())
        else: parts.append(_mr_like(n, seed=i))
    data = b"".join(parts)
    offs = np.cumsum([0] + sizes)
    prm = _lib.make_params(w_kib=256, t=1024)
    streams = gpu.compress_chunks(data, offs, prm)
    assert all(len(s) % 4 == 0 and len(s) > 0 for s in streams)
    back = gpu.decompress_chunks(streams, sizes)
    assert [len(b) for b in back] == sizes and b"".join(back) == data
    assert gpu.compress(parts[4], prm) == streams[4]


def test_config5_full_size_mr_like_round_trip_w512_t4096(gpu):
    """config 5: 9 970 564 bytes (size of Silesia 'mr'), 16-bit image-like samples, -w 512 -t 4096: compress + decompress on the GPU."""
    data = _mr_like(9970564)
    stream = gpu.compress(data, _lib.make_params(w_kib=512, t=4096))
    st = gpu.last_stats
    assert len(stream) % 4 == 0 and sum(list(st.events)[:4]) == st.steps
    assert gpu.decompress(stream, len(data)) == data


def test_huge_t_and_zero_window(gpu, oracle):
    """-t far above any count (K = count_0 then) and -w 0 (no candidates at all; the reference reads out of bounds there)."""
    data = synth.english_like(30_000).tobytes()
    for kw in (dict(w_kib=4, t=1_000_000), dict(w_kib=4, t=70_000)):
        oprm = oracle_lib.params(**kw)  # (the faithful selection loops tc = T..1 like backend.c:76: use the oracle's closed-form path here)
        assert gpu.compress(data, _lib.make_params(**kw)) == oracle.compress(data, oprm, via_m=oracle.scan_m(data, oprm))
    z = gpu.compress(data, _lib.make_params(w_kib=0, t=15))
    assert gpu.decompress(z, len(data)) == data and np.all(gpu.scan_m(data, _lib.make_params(w_kib=0, t=15)) == 0)


def test_sub_batching_is_transparent(gpu, monkeypatch):
    """Batches above X3H_BATCH_BYTES are coded as consecutive sub-batches; streams are independent, so nothing may change."""
    data = synth.english_like(300_000).tobytes()
    cuts = [0, 50_000, 50_000, 120_000, 200_000, 299_999, 300_000]
    prm = _lib.make_params(w_kib=8, t=16)
    whole = gpu.compress_chunks(data, cuts, prm)
    monkeypatch.setenv("X3H_BATCH_BYTES", "100000")
    small = _lib.X3Context(0)
    try:
        assert small.compress_chunks(data, cuts, prm) == whole
        assert small.last_stats.steps == gpu.last_stats.steps
    finally:
        small.close()


# ---- alternative schedules of the same computation (single-stream pipelining, fixed-point mode choice) ------------------------
@pytest.fixture()
def gpu_env(monkeypatch):
    made = []

    def make(**env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        ctx = _lib.X3Context(0)
        made.append(ctx)
        return ctx
    yield make
    for c in made:
        c.close()


@pytest.mark.parametrize("name", CASES)
def test_pipelined_stream_equals_reference_golden(gpu_env, golden, name):
    """every golden stream again with the pipelined schedule forced (X3H_PIPE_MIN=1: parse checkpoints, prefix-wise coding stage,
    segment-wise coder on its own HIP stream) and the fixed-point mode choice forced"""
    ctx = gpu_env(X3H_PIPE_MIN="1", X3H_MODES="fixed")
    c = golden[name]
    assert ctx.compress(c["data"], _lib.params_from_args(c["args"])) == c["expect"]


@pytest.mark.parametrize("gap,sub", [("8192", None), ("256", "3")], ids=["one-slice-or-few", "many-small-slices"])
@pytest.mark.parametrize("name", CASES)
def test_sliced_schedule_equals_reference_golden(gpu_env, golden, name, gap, sub):
    """every golden stream through K3 in slices (code4.hip; api.hip run_sliced) forced on small inputs: X3H_SLICE_GAP=256 cuts them into up to eleven slices
    (carried recency order, context lists, pair ordinals, model counters, coder state), X3H_SLICE_SUB=3 spreads a stream's contexts over three wavefronts"""
    env = dict(X3H_SLICED_MIN="1", X3H_SLICE_GAP=gap, X3H_SLICE_MARKS="0.02,0.05,0.10,0.17,0.26,0.36,0.47,0.59,0.72,0.86")   # (a short stream gets one mark by default)
    if sub:
        env["X3H_SLICE_SUB"] = sub
    ctx = gpu_env(**env)
    c = golden[name]
    assert ctx.compress(c["data"], _lib.params_from_args(c["args"])) == c["expect"]
    if len(c["data"]):
        assert ctx.last_stats.pipelined == 2


@pytest.mark.parametrize("env", [dict(X3H_PIPE_MIN="0", X3H_MODES="serial"), dict(X3H_PIPE_MIN="0", X3H_MODES="fixed"),
                                 dict(X3H_PIPE_MIN="1", X3H_MODES="serial"), dict(X3H_PIPE_MIN="1", X3H_SEG_EMIT="1")],
                         ids=["classic-serial", "classic-fixed", "pipelined-serial", "pipelined-emit-behind-every-segment"])
def test_schedules_agree_on_a_long_stream(gpu, gpu_env, env):
    """3 MiB of text, -w 64 -t 256: the default schedule (pipelined, cost-model choice of the mode pass) against the forced others"""
    data = synth.english_like(3 << 20, seed=77).tobytes()
    prm = _lib.make_params(w_kib=64, t=256)
    want = gpu.compress(data, prm)
    assert gpu_env(**env).compress(data, prm) == want


def test_pipelined_batch_of_long_streams(gpu, gpu_env):
    """a batch of a few long ragged streams (pipelined by default) == the same streams coded one by one with the sequential schedule;
    + a batch above X3H_PIPE_STREAMS falls back to stage-after-stage"""
    sizes = [700_000, 0, 1_300_000, 5, 400_000, 2_000_000]
    text = synth.english_like(sum(sizes), seed=99)
    zipf = synth.zipf_bytes(1_300_000)
    off = np.cumsum([0] + sizes).astype(np.uint64)
    data = text.copy()
    data[int(off[2]):int(off[3])] = zipf
    prm = _lib.make_params(w_kib=64, t=256)
    streams = gpu.compress_chunks(data, off, prm)
    assert gpu.last_stats.pipelined in (1, 2)
    seq = gpu_env(X3H_PIPE_MIN="0")
    for i in range(len(sizes)):
        assert streams[i] == seq.compress(data[int(off[i]):int(off[i + 1])].tobytes(), prm), f"stream {i}"
    few = gpu_env(X3H_PIPE_STREAMS="2", X3H_SLICED="0")
    assert few.compress_chunks(data, off, prm) == streams and few.last_stats.pipelined == 0
    assert gpu_env(X3H_SLICED="0").compress_chunks(data, off, prm) == streams   # the prefix-wise pipelined schedule of rounds 1-3


def test_mid_size_batch_on_masked_streams_equals_plain_streams_and_oracle(gpu, gpu_env, oracle):
    """a batch of 4..192 streams of at most 2 MiB takes K3 in slices on three CU-masked HIP streams (coder on the top 32 CUs, parse and features on the others: api.hip
    sliced_masked_setup) with three to five marks by stream length; the same batch with X3H_SLICE_CUMASK=0 (plain streams, the marks of long streams) and stage after
    stage gives the same streams; two of them are compared with the CPU oracle"""
    sizes = [260_000, 140_000, 99_000, 410_000, 0, 180_000, 300_000, 120_000, 98_304, 650_000]
    text = synth.english_like(sum(sizes), seed=41)
    off = np.cumsum([0] + sizes).astype(np.uint64)
    data = text.copy()
    data[int(off[3]):int(off[4])] = synth.zipf_bytes(sizes[3], offset=7 << 20)
    data[int(off[6]):int(off[7])] = synth.mr_like(sizes[6], seed=12)
    prm = _lib.make_params(w_kib=64, t=256)
    mctx = gpu_env(X3H_SLICE_CUMASK="1")
    masked = mctx.compress_chunks(data, off, prm)
    assert mctx.last_stats.pipelined == 2
    plain = gpu_env(X3H_SLICE_CUMASK="0")
    assert plain.compress_chunks(data, off, prm) == masked and plain.last_stats.pipelined == 2
    assert gpu_env(X3H_PIPE_MIN="0").compress_chunks(data, off, prm) == masked
    for i in (2, 7):
        assert masked[i] == oracle.compress(data[int(off[i]):int(off[i + 1])].tobytes(), oracle_lib.params(w_kib=64, t=256)), f"stream {i}"


def test_decode_batch_of_many_streams(gpu):
    """more than 256 streams select the decoder variant with small LDS tables (ten streams per CU); one stream is rich enough to outgrow
    them (> 2048 dictionary elements -> its tables migrate to global memory)"""
    sizes = [3000] * 299 + [400_000]
    rnd = np.random.default_rng(5)
    parts = [synth.english_like(3000, seed=100 + i).tobytes() for i in range(299)]
    parts.append(bytes(rnd.integers(0, 256, 400_000, dtype=np.uint8)))   # random bytes: every step is a new fragment -> thousands of elements
    data = np.frombuffer(b"".join(parts), dtype=np.uint8)
    off = np.cumsum([0] + sizes).astype(np.uint64)
    prm = _lib.make_params(w_kib=8, t=16)
    streams = gpu.compress_chunks(data, off, prm)
    assert gpu.last_stats.dict_elems > 2048 + 299
    back = gpu.decompress_chunks(streams, sizes)
    assert back == parts
    # the same batch size selects the small-block variants of the parse and mode kernels: their streams == the big variants' (one stream at a time)
    for i in (0, 150, 299):
        assert streams[i] == gpu.compress(parts[i], prm), f"stream {i}"


def test_batch_above_512_streams_equals_oracle_and_the_narrow_forms(gpu, gpu_env, oracle):
    """600 ragged streams (text, Zipf, 16-bit samples, zeros; empty ones): from 513 streams on the coder chains run four to a workgroup
    (x3_ac2_wide_compact_kernel); the batch also takes the time-ranged move-to-front and index-model kernels, the context kernel with ~512-hit ranges on one XCD and
    its own tag gather, compact coder states and LDS-assembled output.  Same bytes as the oracle (sampled) and as the forms they replaced."""
    import oracle_lib
    rng = np.random.default_rng(600)
    parts = []
    for i in range(600):
        n = int(rng.integers(0, 24_000)) if i % 50 else 0
        kind = i % 4
        parts.append(synth.english_like(n, seed=900 + i).tobytes() if kind == 0 else synth.zipf_bytes(n, offset=777 * i).tobytes() if kind == 1
                     else synth.mr_like(n, seed=i).tobytes() if kind == 2 else bytes(n))
    parts[7] = synth.english_like(300_000, seed=5).tobytes()          # one stream of many tiles (and time ranges of several thousand events)
    data = np.frombuffer(b"".join(parts), dtype=np.uint8)
    off = np.cumsum([0] + [len(p) for p in parts]).astype(np.uint64)
    prm = _lib.make_params(w_kib=64, t=256)
    got = gpu.compress_chunks(data, off, prm)
    for i in (0, 1, 2, 3, 7, 50, 301, 598, 599):
        assert got[i] == oracle.compress(parts[i], oracle_lib.params(w_kib=64, t=256)), f"stream {i}"
    old = gpu_env(X3H_AC2_WIDE="0", X3H_MTF_PAR="0", X3H_IDX_PAR="0", X3H_CTX_GATHER="0", X3H_CTX_SUB="4", X3H_CTX_XCD="0").compress_chunks(data, off, prm)
    assert got == old
    assert gpu.decompress_chunks(got, [len(p) for p in parts]) == parts


# ---- chunking behind the C boundary: container, several handles, sub-batches cut on the padded layout ---------------------------------
def test_container_from_hip_streams(gpu, oracle):
    """HIP streams -> X3C1 container (x3h_compress_container) -> unpack -> every chunk == the oracle's stream of that chunk ->
    x3h_decompress_container gives the input back; two handles on GPU 0 working side by side (one host thread each)."""
    from x3_compressor_amd import container
    data = synth.english_like(110_000, seed=3).tobytes()
    kw = dict(w_kib=8, t=16)
    prm = _lib.make_params(**kw)
    with _lib.X3Context(0) as second:
        blob = _lib.compress_container([gpu, second], data, prm, 20_000)
        params, chunks = container.unpack(blob)
        assert params["window_bytes"] == 8192 and len(chunks) == 6
        for i, (raw, s) in enumerate(chunks):
            part = data[i * 20_000:(i + 1) * 20_000]
            assert raw == len(part) and s == oracle.compress(part, oracle_lib.params(**kw)), f"chunk {i}"
        assert _lib.decompress_container([gpu, second], blob, len(data)) == data
        assert _lib.decompress_container([gpu], blob, len(data)) == data
        # the same chunks through the multi-handle batch entry, ragged split
        off = np.array([0, 20_000, 40_000, 40_000, 60_000, 110_000], dtype=np.uint64)
        got = _lib.compress_chunks_multi([gpu, second], np.frombuffer(data, np.uint8), off, prm)
        assert got[0] == chunks[0][1] and got[1] == chunks[1][1] and got[3] == chunks[2][1]
        assert got[2] == oracle.compress(b"", oracle_lib.params(**kw))
    with pytest.raises(_lib.X3Error) as e:
        _lib.decompress_container([gpu], blob[:-4], len(data))
    assert e.value.status == -4
    with pytest.raises(_lib.X3Error) as e:
        _lib.decompress_container([gpu], blob, len(data) - 1)
    assert e.value.status == -3


def test_sub_batch_size_set_per_handle(gpu):
    """x3h_ctx_set_batch_bytes: the chunks of a batch coded (and decoded) a sub-batch at a time -- what the CLI does to keep a short-lived process's
    workspace small -- give the same streams as one batch; values below 1 MiB are refused"""
    data = synth.english_like(3 << 20, seed=21)
    off = np.arange(0, (3 << 20) + 1, 96 << 10, dtype=np.uint64)
    prm = _lib.make_params(w_kib=16, t=32)
    want = gpu.compress_chunks(data, off, prm)
    with _lib.X3Context(0) as small:
        assert small.lib.x3h_ctx_set_batch_bytes(small._h, 1 << 20) == 0
        assert small.lib.x3h_ctx_set_batch_bytes(small._h, 1000) == -1
        got = small.compress_chunks(data, off, prm)
        assert got == want
        back = small.decompress_chunks(got, [int(off[i + 1] - off[i]) for i in range(len(off) - 1)])
        assert b"".join(back) == data.tobytes()


def test_container_through_rccl_equals_host_staged_container(gpu):
    """x3h_compress_container_rccl: streams stay in HBM, one ncclSend/ncclRecv group concatenates them behind the header on the root GPU.
    On a one-GPU machine the block is sent to itself through RCCL; the bytes must equal x3h_compress_container's; two handles on ONE GPU are
    refused (one rank per GPU); a one-chunk input stays the raw stream."""
    data = synth.english_like(300_000, seed=9).tobytes() + synth.zipf_bytes(100_000, offset=5 << 20).tobytes()
    prm = _lib.make_params(w_kib=8, t=16)
    want = _lib.compress_container([gpu], data, prm, 32_768)
    for _ in range(2):  # the second call reuses the cached communicator
        assert _lib.compress_container([gpu], data, prm, 32_768, rccl=True) == want
    assert _lib.compress_container([gpu], data[:30_000], prm, 32_768, rccl=True) == gpu.compress(data[:30_000], prm)
    assert _lib.decompress_container([gpu], want, len(data)) == data
    with _lib.X3Context(0) as second:
        with pytest.raises(_lib.X3Error) as e:
            _lib.compress_container([gpu, second], data, prm, 32_768, rccl=True)
        assert e.value.status == -1
    gpu.lib.x3h_rccl_release()
    assert _lib.compress_container([gpu], data, prm, 65_536, rccl=True) == _lib.compress_container([gpu], data, prm, 65_536)
    gpu.lib.x3h_rccl_release()


def test_sub_batches_are_cut_on_the_padded_layout(gpu, gpu_env):
    """many small chunks under a 512 KiB window: every chunk occupies len + W + slack in the padded layout that K1 indexes with 32 bits,
    so the sub-batch cut must look at that, not at the input bytes (here the limit is lowered to three padded chunks)"""
    data = synth.english_like(40 * 3000, seed=8)
    off = np.arange(0, 40 * 3000 + 1, 3000, dtype=np.uint64)
    prm = _lib.make_params(w_kib=512, t=4096)
    whole = gpu.compress_chunks(data, off, prm)
    small = gpu_env(X3H_BATCH_PAD_BYTES=str(3 * ((512 << 10) + 3000 + 4096 + 256)))
    assert small.compress_chunks(data, off, prm) == whole
    assert small.last_stats.steps == gpu.last_stats.steps
    assert whole[7] == gpu.compress(data[7 * 3000:8 * 3000], prm)


# ---- per-stream feature kernels (code3.hip) forced on inputs that would otherwise take the chip-wide passes ------------------------
@pytest.mark.parametrize("name", CASES)
def test_stream_kernels_equal_reference_golden(gpu_env, golden, name):
    """every golden stream with the per-stream LDS kernels forced (X3H_STREAM_KERNELS=1; they are the default only for >= 48 streams)"""
    ctx = gpu_env(X3H_STREAM_KERNELS="1", X3H_PIPE_MIN="0")
    c = golden[name]
    assert ctx.compress(c["data"], _lib.params_from_args(c["args"])) == c["expect"]


def test_stream_kernels_agree_with_chipwide_passes(gpu_env):
    """a ragged batch of 70 streams (text, Zipf, 16-bit samples, zeros, random; an empty one) coded with the per-stream kernels (default for
    this many streams) and with the chip-wide sort / partition passes: same bytes; plus a 3 MiB stream forced through the stream kernels"""
    rng = np.random.default_rng(17)
    parts = []
    for i in range(70):
        n = int(rng.integers(1, 60_000))
        kind = i % 5
        if kind == 0: parts.append(synth.english_like(n, seed=300 + i).tobytes())
        elif kind == 1: parts.append(synth.zipf_bytes(n, offset=1000 * i).tobytes())
        elif kind == 2: parts.append(synth.mr_like(n, seed=i).tobytes())
        elif kind == 3: parts.append(bytes(n))
        else: parts.append(rng.integers(0, 256, n, dtype=np.uint8).tobytes())
    parts[13] = b""
    data = np.frombuffer(b"".join(parts), dtype=np.uint8)
    off = np.cumsum([0] + [len(p) for p in parts]).astype(np.uint64)
    prm = _lib.make_params(w_kib=64, t=256)
    new = gpu_env(X3H_STREAM_KERNELS="1").compress_chunks(data, off, prm)
    old = gpu_env(X3H_STREAM_KERNELS="0").compress_chunks(data, off, prm)
    assert new == old
    assert gpu_env(X3H_STREAM_KERNELS="1", X3H_ARRANGE="1").compress_chunks(data, off, prm) == old  # hits arranged by one workgroup per stream (x3_arrange_kernel)
    assert gpu_env(X3H_AC2_WIDE="1").compress_chunks(data, off, prm) == old   # coder chains four to a workgroup (default from 513 streams), ragged last workgroup
    big = synth.english_like(3 << 20, seed=77).tobytes()
    assert gpu_env(X3H_STREAM_KERNELS="1", X3H_PIPE_MIN="0").compress(big, prm) == gpu_env(X3H_STREAM_KERNELS="0", X3H_PIPE_MIN="0").compress(big, prm)


# ---- the coder recurrence alone (x3_ac2_kernel's scalar-unit chain) against the oracle's per-symbol intervals ------------------------------
@pytest.mark.parametrize("n,maxtot,seed", [(1, 5, 1), (7, 300, 2), (8, 300, 3), (9, 4000, 4), (33, 2 ** 20, 5), (4096, 2 ** 27, 6),
                                           (100_003, 2 ** 16, 7), (1_000_001, 2 ** 27, 8)])
def test_coder_chain_states_equal_oracle_intervals(gpu, oracle, n, maxtot, seed):
    """ac.c:46-85: random symbol sequences (totals up to 2^27: the magic-multiply division, the closed-form renormalisation s = clz(D)-1-carry,
    the unreduced lo of the asm chain) -- every stored state (mLow, range) and the final mLow must equal the reference arithmetic's, computed
    by the oracle's plain E1/E2/E3 loops.  Sizes around the group-of-8 / 32-symbol-trip boundaries of the kernel."""
    rng = np.random.default_rng(seed)
    total = rng.integers(2, maxtot, n, dtype=np.int64)
    # a mix of near-certain, tiny-probability and uniform symbols
    kind = rng.integers(0, 3, n)
    freq = np.where(kind == 0, np.maximum(total - rng.integers(0, 3, n), 1), np.where(kind == 1, 1, rng.integers(1, total + 1)))
    freq = np.minimum(freq, total)
    cum = (rng.random(n) * (total - freq + 1)).astype(np.int64)
    cum = np.minimum(cum, total - freq)
    states, fin = gpu.coder_chain(cum, freq, total)
    lo, hi = oracle.ac_chain(cum, freq, total)
    g = np.arange(1, (n + 7) // 8)          # state g = the interval after symbol 8g - 1
    assert states[0, 0] == 0 and states[0, 1] == 0x80000000
    assert np.array_equal(states[1:, 0], lo[8 * g - 1])
    assert np.array_equal(states[1:, 1].astype(np.int64), hi[8 * g - 1].astype(np.int64) - lo[8 * g - 1].astype(np.int64) + 1)
    assert fin == int(lo[-1])


# ---- K1 on data with dense classes (more than 2048 members of a class inside a window) and zero runs at the chunk end ----------------------------
def test_scan_dense_classes_equal_brute_force_kernel(gpu, monkeypatch):
    """zero runs, sparse 16-bit samples, periodic data, Zipf bytes ending in zeros: the refinement path of scan2.hip (instead of the
    per-position sweep) and the analytic count of padding members, against the brute-force window sweep of scan.hip; batch of chunks too"""
    monkeypatch.setenv("X3H_SCAN_V1", "1")
    brute = _lib.X3Context(0)
    monkeypatch.delenv("X3H_SCAN_V1")
    rng = np.random.default_rng(33)
    sparse = np.zeros(200_000, np.uint8)
    sparse[rng.integers(0, 200_000, 9000)] = rng.integers(1, 4, 9000)
    cases = [
        (synth.mr_like(300_000).tobytes(), dict(w_kib=64, t=256)),
        (sparse.tobytes(), dict(w_kib=64, t=256)),
        (bytes(150_000), dict(w_kib=64, t=256)),
        (synth.zipf_bytes(200_000).tobytes() + bytes(5000), dict(w_kib=64, t=256)),
        (synth.english_like(100_000).tobytes() + bytes(70_000) + b"tail", dict(w_kib=32, t=100)),
        ((b"\0\0\0\1" * 60_000), dict(w_kib=16, t=40)),
        (synth.mr_like(400_000, seed=3).tobytes(), dict(w_kib=256, t=1024)),
    ]
    try:
        for data, kw in cases:
            prm = _lib.make_params(**kw)
            a, b = gpu.scan_m(data, prm), brute.scan_m(data, prm)
            assert np.array_equal(a, b), f"{kw}: first diff at {first_diff(a, b)}"
    finally:
        brute.close()
    # as chunks of one batch (every chunk has its own padding), against chunk-by-chunk coding
    parts = [c[0][:60_000] for c in cases[:5]]
    data = np.frombuffer(b"".join(parts), dtype=np.uint8)
    off = np.cumsum([0] + [len(p) for p in parts]).astype(np.uint64)
    prm = _lib.make_params(w_kib=64, t=256)
    streams = gpu.compress_chunks(data, off, prm)
    for i, p in enumerate(parts):
        assert streams[i] == gpu.compress(p, prm), f"chunk {i}"
        assert gpu.decompress(streams[i], len(p)) == p


@pytest.mark.parametrize("env", [dict(X3H_SEG_MIN="1"), dict(X3H_SEG_MIN="1", X3H_SEG_SMALL_MAX="0"), dict(X3H_SEG_MIN="1", X3H_SEG_REFINE="0"),
                                 dict(X3H_SEG_MIN="1", X3H_WALK_DENSE="48")],
                         ids=["counters-in-lds", "counters-in-global-memory", "chip-wide-refinement", "dense-from-48-members"])
def test_per_chunk_scan_equals_brute_force_kernel(gpu_env, monkeypatch, env):
    """scan3.hip (K1 of many-chunk batches: one workgroup sorts, level-tests and -- dense classes -- refines one chunk) forced on single
    chunks, against the brute-force window sweep of scan.hip: text, Zipf bytes (exact K for most positions), zero runs and padding counts,
    sparse samples, periodic data, a four-letter alphabet at -t 1 (thousands of classes: three 9-bit passes in the refinement), windows
    larger than the chunk, a chunk of exactly 256 KiB"""
    monkeypatch.setenv("X3H_SCAN_V1", "1")
    brute = _lib.X3Context(0)
    monkeypatch.delenv("X3H_SCAN_V1")
    rng = np.random.default_rng(44)
    sparse = np.zeros(150_000, np.uint8)
    sparse[rng.integers(0, 150_000, 7000)] = rng.integers(1, 4, 7000)
    cases = [
        (synth.english_like(120_000, seed=4).tobytes(), dict(w_kib=64, t=256)),
        (synth.zipf_bytes(90_000, offset=12345).tobytes(), dict(w_kib=64, t=256)),
        (synth.mr_like(262_144, seed=9).tobytes(), dict(w_kib=64, t=256)),
        (sparse.tobytes(), dict(w_kib=32, t=64)),
        (bytes(70_000), dict(w_kib=64, t=256)),
        (synth.english_like(50_000).tobytes() + bytes(40_000) + b"tail", dict(w_kib=32, t=100)),
        ((b"\0\0\0\1" * 30_000), dict(w_kib=16, t=40)),
        (rng.integers(0, 4, 100_000, dtype=np.uint8).tobytes(), dict(w_kib=8, t=1)),
        (rng.integers(0, 4, 60_000, dtype=np.uint8).tobytes(), dict(w_kib=64, t=3)),
        (synth.english_like(30_000, seed=6).tobytes(), dict(w_kib=256, t=1024)),
        (b"abc", dict(w_kib=8, t=16)),
        (synth.english_like(40_000, seed=8).tobytes(), dict(w_kib=64, t=5000)),  # T + 1 beyond a tile: every look-ahead entry comes from memory
        (synth.zipf_bytes(20_000, offset=5).tobytes(), dict(w_kib=16, t=4094)),    # ... and exactly one tile
    ]
    ctx = gpu_env(**env)
    try:
        for data, kw in cases:
            prm = _lib.make_params(**kw)
            a, b = ctx.scan_m(data, prm), brute.scan_m(data, prm)
            assert np.array_equal(a, b), f"{kw}, {len(data)} bytes: first diff at {first_diff(a, b)}"
    finally:
        brute.close()


def test_per_stream_sort_of_the_arrangements_equals_the_chip_wide_sort_and_oracle(gpu_env, oracle):
    """x3_segsort_kernel (code3.hip: a stream's hits grouped by context by the stream's own workgroup; default from 128 streams on) against the chip-wide
    radix sort it replaces and the oracle: 130 ragged streams, among them an empty one, one byte, one long stream (dictionary and tag pairs of
    several thousand: two passes), and every stream again with three passes forced"""
    rng = np.random.default_rng(77)
    parts = []
    for i in range(130):
        n = int(rng.integers(2_000, 30_000))
        parts.append(synth.english_like(n, seed=100 + i).tobytes() if i % 3 else synth.zipf_bytes(n, offset=17 * i).tobytes())
    parts[5], parts[9], parts[40] = b"", b"q", synth.english_like(700_000, seed=3).tobytes()
    data = np.frombuffer(b"".join(parts), np.uint8)
    off = np.cumsum([0] + [len(p) for p in parts]).astype(np.uint64)
    prm = _lib.make_params(w_kib=64, t=256)
    by_default = gpu_env().compress_chunks(data, off, prm)  # (the switches are read when a batch runs: the default first, before any is set)
    want = gpu_env(X3H_SEGSORT="0").compress_chunks(data, off, prm)
    three = gpu_env(X3H_SEGSORT="1", X3H_SEGSORT_PASSES="3").compress_chunks(data, off, prm)
    made_keys = gpu_env(X3H_SEGSORT_GEN="1").compress_chunks(data, off, prm)  # the context0 groups made by the sort itself instead of the element-wise pass
    nine = gpu_env(X3H_SEGSORT_PASSES="1", X3H_SEGSORT_GEN="0", X3H_SEGSORT_NINE="1").compress_chunks(data, off, prm)  # one pass of 9-bit digits wherever the keys fit
    for i in range(len(parts)):
        assert by_default[i] == want[i], f"per-stream sort: stream {i}"
        assert three[i] == want[i], f"per-stream sort, three passes: stream {i}"
        assert made_keys[i] == want[i], f"per-stream sort making its keys: stream {i}"
        assert nine[i] == want[i], f"per-stream sort, nine-bit digits: stream {i}"
    for i in (0, 1, 5, 9):
        assert want[i] == oracle.compress(parts[i], oracle_lib.params(w_kib=64, t=256)), f"stream {i} against the oracle"


def test_many_streams_with_one_oversized_dictionary(gpu, oracle):
    """50 streams, one of which has more dictionary elements (> 8192) than the LDS tables of the per-stream kernels hold: the whole batch
    takes the chip-wide passes instead (x3_code_v2_run derives the token prefix sums itself in that case) -- same bytes as stream by stream"""
    rng = np.random.default_rng(91)
    words = rng.integers(0, 256, (9000, 5), dtype=np.uint8)
    big = np.repeat(words, 4, axis=0).reshape(-1).tobytes()          # every 5-byte word four times in a row: ~9000 elements at -w 1 -t 1
    parts = [synth.english_like(2000 + 37 * i, seed=500 + i).tobytes() for i in range(49)] + [big]
    data = np.frombuffer(b"".join(parts), dtype=np.uint8)
    off = np.cumsum([0] + [len(p) for p in parts]).astype(np.uint64)
    kw = dict(w_kib=1, t=1)
    prm = _lib.make_params(**kw)
    streams = gpu.compress_chunks(data, off, prm)
    assert gpu.last_stats.dict_elems > 8192
    for i in (0, 17, 48, 49):
        assert streams[i] == gpu.compress(parts[i], prm), f"stream {i}"
    assert streams[3] == oracle.compress(parts[3], oracle_lib.params(**kw))
    assert gpu.decompress(streams[49], len(big)) == big


def test_device_resident_batch_round_trip(gpu, oracle):
    """x3h_compress_chunks_dev -> streams compacted back to back in HBM -> x3h_decompress_chunks_dev: nothing crosses PCIe but lengths;
    streams equal the oracle's, decoded bytes equal the input, capacities exact and generous"""
    rng = np.random.default_rng(11)
    parts = [synth.english_like(5000, seed=1).tobytes(), b"", synth.zipf_bytes(3000, offset=9).tobytes(), bytes(2500), b"q",
             rng.integers(0, 256, 1800, dtype=np.uint8).tobytes(), synth.mr_like(4000, seed=3).tobytes()]
    kw = dict(w_kib=2, t=4)
    sizes = [len(q) for q in parts]
    off = np.cumsum([0] + sizes).astype(np.uint64)
    dev = torch.device("cuda", 0)
    d_in = torch.from_numpy(np.frombuffer(b"".join(parts), dtype=np.uint8).copy()).to(dev)
    stride = 8192
    d_out = torch.zeros(stride * len(parts), dtype=torch.uint8, device=dev)
    lens, _ = gpu.compress_chunks_dev(d_in.data_ptr(), off, _lib.make_params(**kw), d_out.data_ptr(), stride)
    want = [oracle.compress(q, oracle_lib.params(**kw)) for q in parts]
    for i, w in enumerate(want):
        assert d_out[i * stride:i * stride + int(lens[i])].cpu().numpy().tobytes() == w, f"stream {i}"
    d_cmp = torch.cat([d_out[i * stride:i * stride + int(lens[i])] for i in range(len(parts))])   # back to back: every offset a multiple of 4
    ioff = np.cumsum([0] + [int(x) for x in lens]).astype(np.uint64)
    caps = [n + (0 if i % 2 else 5) for i, n in enumerate(sizes)]
    ooff = np.cumsum([0] + caps).astype(np.uint64)
    d_back = torch.full((int(ooff[-1]) + 1,), 0xEE, dtype=torch.uint8, device=dev)
    dlens, _ = gpu.decompress_chunks_dev(d_cmp.data_ptr(), ioff, d_back.data_ptr(), ooff)
    assert [int(x) for x in dlens] == sizes
    hb = d_back.cpu().numpy()
    for i, q in enumerate(parts):
        assert hb[int(ooff[i]):int(ooff[i]) + sizes[i]].tobytes() == q, f"decoded stream {i}"
    assert hb[-1] == 0xEE                                                          # nothing written past the last capacity
    with pytest.raises(_lib.X3Error) as e:                                         # an offset that is not a multiple of 4
        gpu.decompress_chunks_dev(d_cmp.data_ptr(), np.array([2, int(ioff[1])], dtype=np.uint64), d_back.data_ptr(), ooff[:2])
    assert e.value.status == -1


@pytest.mark.parametrize("nstreams,nbytes,vocab,floor", [(1, 1_500_000, 12_000, 8192), (300, 150_000, 5_000, 3584), (1100, 40_000, 2_500, 1792), (2100, 20_000, 1_000, 768)],
                         ids=["big_tables_8192", "middle_tables_3584", "eight_per_cu_tables_1792", "small_tables_768"])
def test_decoder_tables_migrate_out_of_lds(gpu, nstreams, nbytes, vocab, floor):
    """dictionaries that outgrow the decoder's LDS tables in each of its four kernel variants (random words of a large vocabulary, window =
    the whole stream, -t 1: every word becomes an element -- ranks beyond 64, context lists beyond 64 items): the tables continue in
    global memory mid-stream and the round trip is exact"""
    rng = np.random.default_rng(77 + nstreams)
    words = np.concatenate([rng.integers(97, 123, (vocab, 6), dtype=np.uint8), np.full((vocab, 1), 32, np.uint8)], axis=1)
    data = np.ascontiguousarray(words[rng.integers(0, vocab, nstreams * nbytes // 7 + 1)].reshape(-1)[:nstreams * nbytes])
    parts = [data[i * nbytes:(i + 1) * nbytes].tobytes() for i in range(nstreams)]
    off = np.arange(0, (nstreams + 1) * nbytes, nbytes, dtype=np.uint64)
    streams = gpu.compress_chunks(data, off, _lib.make_params(w_kib=2048, t=1))
    back = gpu.decompress_chunks(streams, [nbytes + (i % 3) for i in range(nstreams)])
    st = gpu.last_stats
    assert st.dict_elems / nstreams > floor, f"only {st.dict_elems / nstreams:.0f} elements per stream, the spill path was not taken"
    assert back == parts


def test_decoder_division_is_exact(tmp_path):
    """range / total of the decoder's chain (dec_div, decode.hip: v_rcp_f64 + one Newton step aimed 2^-43 low + a one-sided fix-up) against the device's own integer
    division on ~1.7 M operand pairs chosen against a reciprocal: exact multiples of the divisor and their neighbours, divisors 1..4096, powers of two +-2, random
    ones up to 2^28, dividends up to 2^31 (tests/hip/div_check.hip, built here with hipcc)"""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this box")
    exe = tmp_path / "div_check"
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hip", "div_check.hip")
    subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-o", str(exe), src], check=True, capture_output=True, timeout=600)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and " wrong 0" in r.stdout, r.stdout + r.stderr
