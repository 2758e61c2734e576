"""The REAL shapes of BASELINE.json's configs pinned to the REAL reference: tests/golden/manifest_sha.json holds the sha256 of the
streams oracle/_ref/x3 (compiled from /root/reference) wrote for inputs that synth.py regenerates -- config 2 at full size
(10 192 446 bytes, millions of steps), configs 3 and 5 with the window SMALLER than the input (the window edge of backend.c:60-74 at
W = 262 144 / 524 288), whole 8 MiB chunks of config 4 (incl. the last one), and since round 3: ALL TWELVE config-3 streams at the
full Silesia sizes, config 5 at full size, and streams with more than 2^24 parse steps (where `(float)total` of x3.c:152-172 starts to
round).  Every case runs under the default schedule and with pipelining off; the long ones are also decoded back on the GPU, and the
ones past 2^24 steps run under both forms of the mode choice (X3H_MODES=serial|fixed)."""
import hashlib
import json
import os

import pytest
import torch  # before libx3hip.so is loaded: torch brings its own HIP runtime, and the process must end up with one (the first one loaded)

import golden_util
from x3_compressor_amd import _lib, synth

pytestmark = pytest.mark.gpu
MAN = json.load(open(os.path.join(golden_util.HERE, "manifest_sha.json")))
PIECES = sorted(n for n in MAN if MAN[n]["generator"] == "piece")   # chunks of the batches bench.py times (round 5): tested in their batch, below
# ONE stream of X3H_MAX_CHUNK = 2^28 - 4096 bytes (round 5): text (1.2e8 parse steps) and Zipf bytes (nearly a step per byte: the event model's total reaches 2^28 -- hours of
# reference time, in the manifest once tests/golden/make_golden_sha.py big_zipf_maxchunk has finished); their own test below (a decode of 1e8 steps would take minutes)
MAXCHUNK = ("big_english_maxchunk_w1_t4", "big_zipf_maxchunk_w1_t4")
CASES = sorted(n for n in MAN if not n.startswith("cli_") and n not in PIECES and n not in MAXCHUNK)
LONG = [n for n in CASES if n.startswith(("big_", "cfg3_full_", "cfg5_full_"))]   # minutes to hours of reference time each
PAST_2_24 = [n for n in LONG if MAN[n]["input_len"] >= (24 << 20)]   # 0.5-0.8 parse steps per byte: more than 2^24 steps (asserted below)


@pytest.fixture(scope="module")
def inputs():
    cache = {}

    def get(name):
        if name not in cache:
            e = MAN[name]
            data = getattr(synth, e["generator"])(**e["generator_args"])
            assert hashlib.sha256(data.tobytes()).hexdigest() == e["input_sha256"], f"generator output of {name} drifted"
            cache.clear()  # the inputs are large: keep one
            cache[name] = data
        return cache[name]
    return get


SCHEDULES = {"default-schedule": {}, "stage-after-stage": {"X3H_PIPE_MIN": "0"}, "prefix-wise-pipelined": {"X3H_SLICED": "0"}}
# every case under the default schedule (K3 in slices for these sizes) and stage after stage; the prefix-wise pipelined schedule of rounds 1-3 (what a
# dictionary beyond the sliced kernels' tables falls back to) on the cases that take seconds, not minutes
RUNS = [(n, s) for n in CASES for s in ("default-schedule", "stage-after-stage")] + [(n, "prefix-wise-pipelined") for n in CASES if n not in LONG]


@pytest.mark.parametrize("name,sched", RUNS, ids=[f"{s}-{n}" for n, s in RUNS])
def test_stream_sha_equals_reference(monkeypatch, inputs, name, sched):
    env = SCHEDULES[sched]
    e = MAN[name]
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    with _lib.X3Context(0) as ctx:
        got = ctx.compress(inputs(name), _lib.params_from_args(e["args"]))
        st = ctx.last_stats
    assert len(got) == e["output_len"], f"{name}: {len(got)} bytes, the reference wrote {e['output_len']}"
    assert hashlib.sha256(got).hexdigest() == e["output_sha256"]
    if not env and e["input_len"] >= (256 << 10):
        assert st.pipelined in (1, 2)   # an overlapped schedule: 2 = in slices with carried state (code4.hip), 1 = prefix-wise (run_pipelined)
    if env.get("X3H_PIPE_MIN") == "0":
        assert st.pipelined == 0
    if env.get("X3H_SLICED") == "0" and e["input_len"] >= (256 << 10):
        assert st.pipelined == 1
    if name in PAST_2_24:
        # model_events.total = 2051 + steps is no longer exact as a float here (x3.c:152-172,236-244; ac.c:108-113)
        assert st.steps > (1 << 24), st.steps


@pytest.mark.parametrize("modes", ["serial", "fixed"])
@pytest.mark.parametrize("name", PAST_2_24)
def test_mode_choice_past_2_24_steps(monkeypatch, inputs, name, modes):
    """both forms of the mode choice (serial kernel / chip-wide fixed point) on the streams whose event totals exceed 2^24"""
    e = MAN[name]
    monkeypatch.setenv("X3H_MODES", modes)
    with _lib.X3Context(0) as ctx:
        got = ctx.compress(inputs(name), _lib.params_from_args(e["args"]))
    assert len(got) == e["output_len"] and hashlib.sha256(got).hexdigest() == e["output_sha256"]


def test_long_streams_decode_back(inputs):
    """every long reference-pinned stream decoded back ON THE GPU (one batch per window setting: the streams decode side by side): the 24-/27-bit
    packings of the decoder (position << 5 | len-1, jump tables) at 33-51 MB"""
    by_args = {}
    for n in LONG:
        by_args.setdefault(tuple(MAN[n]["args"]), []).append(n)
    for args, names in by_args.items():
        streams, datas = [], []
        with _lib.X3Context(0) as ctx:
            for n in names:
                d = inputs(n).tobytes()
                s = ctx.compress(d, _lib.params_from_args(list(args)))
                assert hashlib.sha256(s).hexdigest() == MAN[n]["output_sha256"], n
                streams.append(s); datas.append(d)
            back = ctx.decompress_chunks(streams, [len(d) for d in datas])
        for n, d, b in zip(names, datas, back):
            assert b == d, n


@pytest.mark.parametrize("name", MAXCHUNK)
def test_longest_single_stream_equals_reference(name):
    """the reference codes any input as ONE stream (x3.c:577-611); the library's longest stream is X3H_MAX_CHUNK = 2^28 - 4096 bytes (include/x3hip.h has the bound: every
    model total stays < 2^28, so a coder step is never a single value).  One stream of exactly that size -- 1.2e8 parse steps -- against the real reference's `x3 -z -w 1 -t 4`; one byte more is refused."""
    if name not in MAN:
        pytest.skip(f"the reference stream of {name} is not in the manifest yet (tests/golden/make_golden_sha.py {name}: hours of reference time)")
    e = MAN[name]
    assert e["input_len"] == (1 << 28) - 4096
    data = golden_util.sha_input(e)
    with _lib.X3Context(0) as ctx:
        got = ctx.compress(data, _lib.params_from_args(e["args"]))
        st = ctx.last_stats
        assert st.steps > 100_000_000
        assert golden_util.pinned_stream_ok(e, got), f"{len(got)} bytes, the reference wrote {e['output_len']}"
        import numpy as np
        with pytest.raises(_lib.X3Error) as err:
            ctx.compress(np.zeros((1 << 28) - 4095, dtype=np.uint8), _lib.params_from_args(e["args"]))
        assert err.value.status == -1


PINNED4 = (0, 1, 15, 64, 127)   # chunks of config 4's 128 x 8 MiB Zipf stream with a reference stream in the manifest
CHUNK4 = 8 << 20


def _pinned4(chunk):
    return MAN[f"cfg4_zipf_chunk{chunk}_8m_w64_t256"]


def test_config4_share_in_the_form_that_is_timed():
    """BASELINE config 4 as bench.py times it on every GPU: ONE batch of 16 x 8 MiB chunks of the Zipf stream (rank 0's share, chunks 0..15), inputs
    and streams resident in HBM (x3h_compress_chunks_dev), under the DEFAULT schedule -- for this shape K3 in slices with carried state (code4.hip, api.hip run_sliced:
    parse | feature stages | coder | bit emission of a stream's slices overlap on four HIP streams).  The pinned chunks 0, 1, 15 must equal the real reference's
    `x3 -z -w 64 -t 256` of that chunk alone (x3.c:372-434,593-611)."""
    import numpy as np
    per = 16
    data = synth.zipf_bytes(per * CHUNK4)
    prm = _lib.make_params(w_kib=64, t=256)
    off = np.arange(0, (per + 1) * CHUNK4, CHUNK4, dtype=np.uint64)
    stride = (CHUNK4 + (CHUNK4 >> 2) + 4096 + 3) & ~3
    dev = torch.device("cuda", 0)
    d_in = torch.from_numpy(data).to(dev)
    d_out = torch.empty(stride * per, dtype=torch.uint8, device=dev)
    with _lib.X3Context(0) as ctx:
        lens, st = ctx.compress_chunks_dev(d_in.data_ptr(), off, prm, d_out.data_ptr(), stride)
        torch.cuda.synchronize()
        assert st.pipelined == 2, "16 x 8 MiB is a few-long-streams batch: the sliced, overlapped schedule is its default"
        for c in (c for c in PINNED4 if c < per):
            e = _pinned4(c)
            got = d_out[c * stride:c * stride + int(lens[c])].cpu().numpy().tobytes()
            assert len(got) == e["output_len"], f"chunk {c}: {len(got)} bytes, the reference wrote {e['output_len']}"
            assert hashlib.sha256(got).hexdigest() == e["output_sha256"], f"chunk {c}"
        # and decoded back as one batch, streams and bytes in HBM
        ioff = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
        d_cmp = torch.cat([d_out[i * stride:i * stride + int(lens[i])] for i in range(per)])
        d_back = torch.empty(per * CHUNK4, dtype=torch.uint8, device=dev)
        dl, _ = ctx.decompress_chunks_dev(d_cmp.data_ptr(), ioff, d_back.data_ptr(), off)
        torch.cuda.synchronize()
        assert int(dl.sum()) == per * CHUNK4 and torch.equal(d_back, d_in)


def _batch_on_gpu(ctx, data, cb, prm):
    """one batch of independent `cb`-byte chunks, device-resident in and out, exactly as bench.py's chunk_batch() runs it -> (streams getter, stats)"""
    import numpy as np
    total = int(data.size)
    off = np.array(list(range(0, total, cb)) + [total], dtype=np.uint64)
    stride = (cb + (cb >> 1) + 4096 + 3) & ~3
    dev = torch.device("cuda", 0)
    d_in = torch.from_numpy(data).to(dev)
    d_out = torch.empty(stride * (len(off) - 1), dtype=torch.uint8, device=dev)
    lens, st = ctx.compress_chunks_dev(d_in.data_ptr(), off, prm, d_out.data_ptr(), stride)
    torch.cuda.synchronize()
    return (lambda c: d_out[c * stride:c * stride + int(lens[c])].cpu().numpy().tobytes()), st, len(off) - 1


def _pieces_of(base, prefix):
    return {n: MAN[n] for n in PIECES if n.startswith(prefix) and MAN[n]["generator_args"]["base"] == base}


MANY_STREAM_SWITCHES = {"default-switches": {}, "chip-wide-sorts": {"X3H_SEGSORT": "0", "X3H_SEG_MIN": "0"}}


@pytest.mark.parametrize("switches", list(MANY_STREAM_SWITCHES))
def test_many_chunks_batch_in_the_form_that_is_timed(monkeypatch, switches):
    """bench.py's `many_chunks_batch`: 1024 x 256 KiB chunks (half English-like text, half Zipf bytes) as ONE device-resident batch under `-w 64 -t 256` --
    the window is smaller than a chunk's padded slot, the per-chunk K1 sort (scan3.hip), the per-stream kernels of code3.hip incl. the one-pass 9-bit arrangement
    sort and four coder chains per workgroup are what runs.  Fourteen chunks (first / middle / last of either half and both sides of the text -> Zipf boundary)
    must equal the real reference's `x3 -z -w 64 -t 256` of that chunk alone (x3.c:372-434,593-611); once more with the chip-wide sorts (X3H_SEGSORT=0 X3H_SEG_MIN=0)."""
    for k, v in MANY_STREAM_SWITCHES[switches].items():
        monkeypatch.setenv(k, v)
    pinned = _pieces_of("many_chunks_mix", "mcb_mix256m_")
    assert len(pinned) >= 12
    data = synth_base("many_chunks_mix")
    with _lib.X3Context(0) as ctx:
        stream_of, st, nch = _batch_on_gpu(ctx, data, synth.MANY_CHUNK_BYTES, _lib.make_params(w_kib=64, t=256))
        assert nch == 1024 and st.pipelined == 0
        for n, e in pinned.items():
            c = e["generator_args"]["start"] // synth.MANY_CHUNK_BYTES
            assert hashlib.sha256(data[c * synth.MANY_CHUNK_BYTES:(c + 1) * synth.MANY_CHUNK_BYTES].tobytes()).hexdigest() == e["input_sha256"], n
            assert golden_util.pinned_stream_ok(e, stream_of(c)), f"{n}: chunk {c} of the timed batch differs from the real reference's stream"


def synth_base(base):
    """one of bench.py's batches (generated once per process, golden_util keeps it)"""
    golden_util.sha_input(dict(generator="piece", generator_args=dict(base=base, start=0, n=0), input_sha256=hashlib.sha256(b"").hexdigest()))
    return golden_util._BASES[base]


def test_dense_class_batch_in_the_form_that_is_timed():
    """bench.py's `many_chunks_dense_classes`: 64 MiB of mr-like 16-bit samples as 256 x 256 KiB chunks (zero runs: the per-chunk refinement of K1, 24-32 dictionary
    lengths hitting at every position of K2); four chunks against the real reference"""
    pinned = _pieces_of("dense_batch", "mcb_dense64m_")
    assert len(pinned) >= 4
    data = synth_base("dense_batch")
    with _lib.X3Context(0) as ctx:
        stream_of, st, nch = _batch_on_gpu(ctx, data, synth.MANY_CHUNK_BYTES, _lib.make_params(w_kib=64, t=256))
        assert nch == 256
        for n, e in pinned.items():
            c = e["generator_args"]["start"] // synth.MANY_CHUNK_BYTES
            assert golden_util.pinned_stream_ok(e, stream_of(c)), f"{n}: chunk {c}"


@pytest.mark.parametrize("nch", [40, 64, 128])
def test_chunked_same_bytes_in_the_form_that_is_timed(nch):
    """bench.py's `chunked_same_bytes`: the dickens-sized text cut into 40 / 64 (K3 in slices on the CU-masked streams) / 128 chunks (per-stream kernels); the first and
    the (shorter) last chunk against the real reference"""
    data = synth_base("english_like")
    cb = (int(data.size) + nch - 1) // nch
    with _lib.X3Context(0) as ctx:
        stream_of, st, got_nch = _batch_on_gpu(ctx, data, cb, _lib.make_params(w_kib=64, t=256))
        assert got_nch == nch
        if nch <= 96:
            assert st.pipelined == 2
        for i in (0, nch - 1):
            e = MAN[f"csb_dickens_{nch}x_chunk{i}_w64_t256"]
            assert (e["generator_args"]["start"], e["generator_args"]["n"]) == synth.same_bytes_chunk_range(nch, i)
            assert golden_util.pinned_stream_ok(e, stream_of(i)), f"{nch} chunks: chunk {i}"


@pytest.mark.parametrize("name", PIECES)
def test_pinned_piece_as_a_stream_of_its_own(name):
    """every pinned chunk once more as ONE stream through x3h_compress (the schedule a lone 80-256 KiB stream takes)"""
    e = MAN[name]
    with _lib.X3Context(0) as ctx:
        got = ctx.compress(golden_util.sha_input(e), _lib.params_from_args(e["args"]))
    assert golden_util.pinned_stream_ok(e, got)


def test_container_rccl_over_every_gpu_of_the_box():
    """x3h_compress_container_rccl with ndevices == x3h_device_count(): every GPU codes its contiguous block of config-4 chunks, ONE RCCL
    send/receive group concatenates the blocks on GPU 0.  Bytes must equal the host-staged container's, and the pinned chunks the real
    reference's streams.  Skipped on a one-GPU box (the one-rank self-send is test_container_through_rccl_equals_host_staged_container)."""
    lib = _lib.load_library()
    ndev = int(lib.x3h_device_count())
    if ndev < 2:
        pytest.skip("one GPU: the multi-rank gather cannot run here")
    ndev = min(ndev, 8)
    per = 2  # chunks per GPU: 16 MiB each, the pinned chunks 0 and 1 are on GPU 0
    data = synth.zipf_bytes(ndev * per * CHUNK4).tobytes()
    prm = _lib.make_params(w_kib=64, t=256)
    ctxs = [_lib.X3Context(d) for d in range(ndev)]
    try:
        got = _lib.compress_container(ctxs, data, prm, CHUNK4, rccl=True)
        want = _lib.compress_container(ctxs, data, prm, CHUNK4)
        assert got == want
        from x3_compressor_amd import container
        _, chunks = container.unpack(got)
        assert len(chunks) == ndev * per
        for c in (c for c in PINNED4 if c < len(chunks)):
            assert hashlib.sha256(chunks[c][1]).hexdigest() == _pinned4(c)["output_sha256"], f"chunk {c}"
        assert _lib.decompress_container(ctxs, got, len(data)) == data
    finally:
        lib.x3h_rccl_release()
        for c in ctxs:
            c.close()


def test_window_smaller_than_input_cases_are_distinct():
    """the small cfg3/cfg5 vectors of manifest.json are the same bytes (16 KiB inside either window); these are not"""
    assert len({MAN[n]["output_sha256"] for n in CASES}) == len(CASES)
    for n in CASES:
        w = int(MAN[n]["args"][MAN[n]["args"].index("-w") + 1]) * 1024
        assert MAN[n]["input_len"] > w, n


def test_real_corpus_files_if_the_box_has_them():
    """X3_CORPUS=DIR: the real Silesia files, pinned by tests/golden/make_golden_sha.py --corpus DIR into tests/golden/manifest_corpus.json with the
    real reference.  Skipped where there is no corpus (this image has none: Silesia is not available offline)."""
    import sys
    cdir = os.environ.get("X3_CORPUS")
    mpath = os.path.join(golden_util.HERE, "manifest_corpus.json")
    if not cdir or not os.path.isdir(cdir) or not os.path.exists(mpath):
        pytest.skip("no corpus directory (X3_CORPUS) or no tests/golden/manifest_corpus.json")
    sys.path.insert(0, golden_util.HERE)
    import make_golden_sha
    man = json.load(open(mpath))
    checked = 0
    with _lib.X3Context(0) as ctx:
        for name, (path, args) in sorted(make_golden_sha.corpus_cases(cdir).items()):
            e = man.get(name)
            data = open(path, "rb").read()
            if e is None or e["input_sha256"] != hashlib.sha256(data).hexdigest() or len(data) > (128 << 20):
                continue
            w, t = int(args[args.index("-w") + 1]), int(args[args.index("-t") + 1])
            s = ctx.compress(data, _lib.make_params(w_kib=w, t=t))
            assert len(s) == e["output_len"] and hashlib.sha256(s).hexdigest() == e["output_sha256"], name
            checked += 1
    assert checked, "a corpus directory and a manifest, but no file of the one is pinned in the other"
