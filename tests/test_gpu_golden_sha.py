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

import golden_util
from x3_compressor_amd import _lib, synth

pytestmark = pytest.mark.gpu
MAN = json.load(open(os.path.join(golden_util.HERE, "manifest_sha.json")))
CASES = sorted(n for n in MAN if not n.startswith("cli_"))
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


@pytest.mark.parametrize("env", [{}, {"X3H_PIPE_MIN": "0"}], ids=["default-schedule", "stage-after-stage"])
@pytest.mark.parametrize("name", CASES)
def test_stream_sha_equals_reference(monkeypatch, inputs, name, env):
    e = MAN[name]
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    with _lib.X3Context(0) as ctx:
        got = ctx.compress(inputs(name), _lib.params_from_args(e["args"]))
        st = ctx.last_stats
    assert len(got) == e["output_len"], f"{name}: {len(got)} bytes, the reference wrote {e['output_len']}"
    assert hashlib.sha256(got).hexdigest() == e["output_sha256"]
    if not env and e["input_len"] >= (256 << 10):
        assert st.pipelined == 1
    if env:
        assert st.pipelined == 0
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


def test_window_smaller_than_input_cases_are_distinct():
    """the small cfg3/cfg5 vectors of manifest.json are the same bytes (16 KiB inside either window); these are not"""
    assert len({MAN[n]["output_sha256"] for n in CASES}) == len(CASES)
    for n in CASES:
        w = int(MAN[n]["args"][MAN[n]["args"].index("-w") + 1]) * 1024
        assert MAN[n]["input_len"] > w, n
