"""Host-side logic that the measurements and the multi-stream path rest on (no GPU needed)."""
import hashlib

import numpy as np
import pytest

from x3_compressor_amd import _lib, container, synth


def test_workload_generators_are_pinned():
    """bench.py, the golden vectors and the GPU tests all draw from these generators: their bytes must never drift."""
    assert hashlib.sha256(synth.english_like(65536).tobytes()).hexdigest() == "a90b8b366c6280e0cfc9afde17a35b1bf7249dff6f46859a633a861f3855a1b9"
    assert hashlib.sha256(synth.zipf_bytes(65536).tobytes()).hexdigest() == "340cdf1af482b4131b0def1123de4fdecef99843bd3c05599dadf6db18861dfe"
    assert hashlib.sha256(synth.zipf_bytes(4096, offset=1 << 20).tobytes()).hexdigest() == "54a60a617014e9e693d07f11d7085e132c720062e0976adb033e53778a26166b"
    assert hashlib.sha256(synth.mr_like(65536).tobytes()).hexdigest() == "a6eddd64c1436dfd350a6a44440b7f29e58018cc7ba61d735d8113fa466cb5fb"
    assert hashlib.sha256(synth.mr_like(4097, seed=9).tobytes()).hexdigest() == "ba6d0fa91053c457edf31f40d1ed454b91cb22e88234202b374d423a5a23b926"
    assert synth.mr_like(300_000).tobytes()[:100_001] == synth.mr_like(100_001).tobytes()  # a prefix of a longer volume
    # config 4: the stream is defined byte by byte, so any offset must splice seamlessly
    a = synth.zipf_bytes(3000)
    assert np.array_equal(np.concatenate([synth.zipf_bytes(1234), synth.zipf_bytes(3000 - 1234, offset=1234)]), a)


def test_zipf_threshold_table_is_the_committed_one():
    """256 cumulative uint32 thresholds for P(r) ~ 1/(r+1), computed from exact rationals."""
    t = synth.ZIPF_THRESHOLDS
    assert len(t) == 256 and int(t[0]) == 701294150 and int(t[1]) == 1051941225 and int(t[254]) == 4292227865 and int(t[255]) == 2**32 - 1
    assert np.all(np.diff(t.astype(np.int64)) > 0)
    z = synth.zipf_bytes(1 << 18)
    p = np.bincount(z, minlength=256) / z.size
    assert abs(-(p[p > 0] * np.log2(p[p > 0])).sum() - 6.2217) < 0.02  # source entropy quoted in SURVEY.md


def test_english_like_looks_like_prose():
    e = synth.english_like(200_000).tobytes()
    assert e[:1].isupper() and b" the " in e and b". " in e and e.count(b" ") > 30_000
    assert max(e) < 128


def test_params_mapping_matches_cli_letters():
    p = _lib.params_from_args(["-w", "64", "-t", "256", "-m", "3", "-n", "2", "-x"])
    assert (p.window_bytes, p.max_match_count, p.factor1, p.factor2, p.nl_mode) == (65536, 256, 3, 2, 1)  # x3.c:499-513
    d = _lib.make_params()
    assert (d.window_bytes, d.max_match_count, d.factor1, d.factor2, d.nl_mode) == (8192, 15, 4, 0, 0)    # backend.c:8,21,33-34
    with pytest.raises(ValueError):
        _lib.params_from_args(["-q"])


def test_container_round_trip_and_errors():
    prm = _lib.make_params(w_kib=64, t=256)
    streams = [b"\x01\x02\x03\x04", b"\xff\x17\x00\x00", b"\xff" * 12]
    blob = container.pack(streams, [10, 0, 99], prm)
    assert blob[:4] == b"X3C1" and len(blob) == 32 + 3 * 16 + 20
    params, chunks = container.unpack(blob)
    assert params == dict(window_bytes=65536, max_match_count=256, factor1=4, factor2=0, nl_mode=0)
    assert chunks == [(10, streams[0]), (0, streams[1]), (99, streams[2])]
    assert container.split_offsets(10, 4) == [0, 4, 8, 10] and container.split_offsets(0, 4) == [0, 0]
    # the byte layout is part of the format (little-endian, x3hip.h): magic, version, params, nchunks, (raw, comp) table
    import struct
    assert struct.unpack_from("<4sIIiIIiI", blob) == (b"X3C1", 1, 65536, 256, 4, 0, 0, 3)
    assert struct.unpack_from("<QQQQQQ", blob, 32) == (10, 4, 0, 4, 99, 12)


def test_container_rejects_malformed_input():
    """unpack() validates every length against the blob (a table that points outside it must not be trusted)."""
    prm = _lib.make_params()
    blob = bytearray(container.pack([b"\x00" * 8, b"\x00" * 4], [100, 50], prm))
    for bad in (bytes(blob) + b"x",                     # trailing bytes
                bytes(blob[:-1]),                        # truncated payload
                bytes(blob[:40]),                        # truncated table
                bytes(blob[:20])):                       # truncated header
        with pytest.raises(ValueError):
            container.unpack(bad)
    import struct
    b2 = bytearray(blob); struct.pack_into("<I", b2, 4, 2)             # unknown version
    b3 = bytearray(blob); struct.pack_into("<I", b3, 28, 0x7FFFFFFF)   # absurd chunk count
    b4 = bytearray(blob); struct.pack_into("<Q", b4, 32 + 8, 1 << 40)  # comp_len beyond the blob
    b5 = bytearray(blob); struct.pack_into("<Q", b5, 32, 1 << 28)      # raw_len above X3H_MAX_CHUNK
    b6 = bytearray(blob); struct.pack_into("<Q", b6, 32 + 8, 6); struct.pack_into("<Q", b6, 48 + 8, 6)  # not whole words
    for bad in (b2, b3, b4, b5, b6):
        with pytest.raises(ValueError):
            container.unpack(bytes(bad))
    # no magic: one raw stream, whatever it is
    assert container.unpack(b"\xff\x17\x00\x00") == (None, [(None, b"\xff\x17\x00\x00")])
    assert container.unpack(b"") == (None, [(None, b"")])
