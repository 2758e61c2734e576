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


def test_emulated_decoder_errors(emu, golden):
    with pytest.raises(_lib.X3Error) as e:
        emu.decompress(golden["zeros5000"]["expect"], 100)   # ratio > 64:1 -- the reference overruns its buffer here (x3.c:621)
    assert e.value.status == -3
    with pytest.raises(_lib.X3Error) as e:
        emu.decompress(b"\x12\x34\x56\x78" * 50, 1000)
    assert e.value.status == -4
