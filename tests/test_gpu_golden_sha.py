"""The REAL shapes of BASELINE.json's configs pinned to the REAL reference: tests/golden/manifest_sha.json holds the sha256 of the
streams oracle/_ref/x3 (compiled from /root/reference) wrote for inputs that synth.py regenerates -- config 2 at full size
(10 192 446 bytes, millions of steps), configs 3 and 5 with the window SMALLER than the input (the window edge of backend.c:60-74 at
W = 262 144 / 524 288), a whole 8 MiB chunk of config 4.  Every case runs under the default schedule and with pipelining off."""
import hashlib
import json
import os

import pytest

import golden_util
from x3_compressor_amd import _lib, synth

pytestmark = pytest.mark.gpu
MAN = json.load(open(os.path.join(golden_util.HERE, "manifest_sha.json")))
CASES = sorted(n for n in MAN if not n.startswith("cli_"))


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


def test_window_smaller_than_input_cases_are_distinct():
    """the small cfg3/cfg5 vectors of manifest.json are the same bytes (16 KiB inside either window); these are not"""
    assert len({MAN[n]["output_sha256"] for n in CASES}) == len(CASES)
    for n in CASES:
        w = int(MAN[n]["args"][MAN[n]["args"].index("-w") + 1]) * 1024
        assert MAN[n]["input_len"] > w, n
