"""The drop-in CLI on the GPU box: csrc/x3 (C, links only include/x3hip.h) run as a program on the reference's own
golden streams and with the reference's file-name / clobber / stdin rules (x3.c:484-548, file.c:47-55), plus the additive
chunk / multi-GPU options and the X3C1 container."""
import hashlib
import json
import os
import subprocess

import pytest

import golden_util
from x3_compressor_amd import container, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
X3 = os.path.join(ROOT, "x3_compressor_amd", "csrc", "x3")
CASES = sorted(golden_util.load_cases().keys())


def run(args, **kw):
    """one x3 process; no retry: a process that dies with a signal fails the test (the CLI prints the phase and the stack of a fatal signal;
    profiles/r03_cli_exit_loops.txt: 2 600 fresh processes, none with a non-zero exit status)"""
    return subprocess.run([X3, *args], capture_output=True, timeout=600, **kw)


@pytest.mark.parametrize("name", CASES)
def test_cli_compress_and_decompress_golden(tmp_path, golden, name):
    """`x3 -z <args> in out` == the reference's stream for the same arguments; `x3 -d` gives the input back."""
    c = golden[name]
    i, o, b = tmp_path / "in", tmp_path / "out.x3", tmp_path / "back"
    i.write_bytes(c["data"])
    r = run(["-z", *c["args"], str(i), str(o)])
    assert r.returncode == 0, r.stderr.decode()
    assert o.read_bytes() == c["expect"]
    r = run(["-d", str(o), str(b)])
    assert r.returncode == 0, r.stderr.decode()
    assert b.read_bytes() == c["data"]


def test_cli_file_naming_and_clobber_rules(tmp_path, golden):
    """x3.c:522-548: one argument -> <in>.x3 / strip the last suffix; file.c:47-55: refuse to overwrite without -f."""
    c = golden["gpl16k_default"]
    f = tmp_path / "text.txt"
    f.write_bytes(c["data"])
    assert run([str(f)]).returncode == 0                                  # -z is the default (x3.c:481)
    z = tmp_path / "text.txt.x3"
    assert z.read_bytes() == c["expect"]
    r = run(["-z", str(f)])                                               # output exists
    assert r.returncode != 0 and b"File already exists" in r.stderr
    assert run(["-z", "-f", str(f)]).returncode == 0 and z.read_bytes() == c["expect"]
    f.unlink()
    assert run(["-d", str(z)]).returncode == 0                            # strips ".x3"
    assert f.read_bytes() == c["data"]
    r = run(["-d", str(z)])
    assert r.returncode != 0 and b"File already exists" in r.stderr
    assert run(["-d", "-k", "-f", str(z)]).returncode == 0 and z.exists()  # -k: the input is kept (always)
    r = run(["-z", str(tmp_path / "missing")])
    assert r.returncode != 0 and b"Cannot open input file" in r.stderr
    assert run(["-q"]).returncode != 0                                    # unknown flag (the reference abort()s, x3.c:515)


def test_cli_stdin_stdout_and_statistics(tmp_path, golden):
    """0 arguments: stdin -> stdout (seekable stdin, file.c:25-28); the integer statistics block of x3.c:669-693."""
    c = golden["records_w8_t16"]
    f = tmp_path / "in"
    f.write_bytes(c["data"])
    with open(f, "rb") as fh:
        r = run(["-z", *c["args"]], stdin=fh)
    assert r.returncode == 0 and r.stdout == c["expect"]
    err = r.stderr.decode()
    assert "Compressing..." in err and "max match count: 16" in err and "forward window: 8192" in err and "magic factor 1: 4" in err
    assert f"input stream size: {len(c['data'])}" in err and "dictionary: hit" in err and "number of events: ctx0" in err
    z = tmp_path / "z"
    z.write_bytes(r.stdout)
    with open(z, "rb") as fh:
        r = run(["-d"], stdin=fh)
    assert r.returncode == 0 and r.stdout == c["data"]
    r = run(["-z", "-g", "0", *c["args"], str(f), str(tmp_path / "g.x3")])
    assert r.returncode == 0 and (tmp_path / "g.x3").read_bytes() == c["expect"]
    assert run(["-z", "-g", "99", str(f), str(tmp_path / "h.x3")]).returncode != 0   # no such device


@pytest.mark.parametrize("name", ["gpl16k_default", "cfg1_english64k_w8_t16", "cfg2_english32k_w64_t256", "cfg4_zipf32k_w64_t256", "records_w8_t16", "zeros5000"])
def test_cli_statistics_block_equals_reference(tmp_path, golden, name):
    """the whole statistics block of x3.c:662-693 -- the float size estimates included (x3h_stats.est_bits: -log2f(prob) summed in single precision
    in coding order, x3.c:43,192-193,253-266) -- against the stderr of the REAL reference (oracle/_ref/x3) on the same input: every number within
    1e-3 relative (the terms differ from glibc's log2f by an ulp now and then), the integer lines character by character."""
    c = golden[name]
    ref = golden_util.reference_stderr(c["data"], c["args"])
    if ref is None:
        pytest.skip("oracle/_ref/x3 not present")
    i = tmp_path / "in"
    i.write_bytes(c["data"])
    r = run(["-z", *c["args"], str(i), str(tmp_path / "out.x3")])
    assert r.returncode == 0, r.stderr.decode()
    mine = r.stderr.decode()
    got, want = golden_util.estimate_lines_of(mine), golden_util.estimate_lines_of(ref)
    assert len(got) == len(want) == 4
    for g, w in zip(got, want):
        gn, wn = golden_util.numbers_of(g), golden_util.numbers_of(w)
        assert len(gn) == len(wn) and all(abs(a - b) <= 1e-3 * max(abs(b), 1.0) for a, b in zip(gn, wn)), (g, w)
    clean = lambda t: [l.replace("\x1b[37;1m", "").replace("\x1b[0m", "") for l in t.splitlines()]
    for prefix in ("input stream size", "dictionary:", "real compression ratio", "number of events", "context entries"):
        assert [l for l in clean(mine) if l.startswith(prefix)] == [l for l in clean(ref) if l.startswith(prefix)], prefix


def test_cli_says_when_it_writes_a_container_the_reference_cannot_read(tmp_path):
    """x3.c:577-611 codes any input as ONE stream; this build's longest stream is X3H_MAX_CHUNK (2^28 - 4096 bytes): a larger input without --chunk-kib
    becomes an X3C1 container, and the CLI must say that the reference's `x3 -d` cannot read it.  (zeros: cheap to code, 257 MiB)"""
    f, z, b = tmp_path / "big", tmp_path / "big.x3", tmp_path / "back"
    n = (257 << 20) + 5
    with open(f, "wb") as fh:
        fh.truncate(n)
    r = run(["-z", "-w", "8", "-t", "16", str(f)])
    assert r.returncode == 0, r.stderr.decode()
    assert b"NOTE" in r.stderr and b"cannot read this output" in r.stderr
    blob = z.read_bytes()
    assert blob[:4] == b"X3C1"
    r = run(["-d", str(z), str(b)])
    assert r.returncode == 0, r.stderr.decode()
    assert os.path.getsize(b) == n and not any(open(b, "rb").read(1 << 20))
    import struct
    assert struct.unpack_from("<Q", blob, 32)[0] == (1 << 28) - 4096   # raw length of the first chunk: the longest single stream
    small = tmp_path / "small"
    small.write_bytes(b"abc" * 1000)
    r = run(["-z", str(small)])
    assert r.returncode == 0 and b"NOTE" not in r.stderr
    # between the old limit (128 MiB) and the new one: ONE raw x3 stream, like the reference writes (x3.c:577-611), no container and no note
    mid, midz, midb = tmp_path / "mid", tmp_path / "mid.x3", tmp_path / "midback"
    nm = (200 << 20) + 3
    with open(mid, "wb") as fh:
        fh.truncate(nm)
    r = run(["-z", "-w", "8", "-t", "16", str(mid)])
    assert r.returncode == 0 and b"NOTE" not in r.stderr, r.stderr.decode()
    assert midz.read_bytes()[:4] != b"X3C1" and os.path.getsize(midz) < 8192
    r = run(["-d", str(midz), str(midb)])
    assert r.returncode == 0, r.stderr.decode()
    assert os.path.getsize(midb) == nm


def test_cli_large_file_round_trip_pinned_to_reference(tmp_path):
    """a 5 MiB file (> 2 MiB of compressed stream: the raw-stream decode path must size and grow its output buffer within
    X3H_MAX_CHUNK); the stream's sha256 is the REAL reference's (tests/golden/manifest_sha.json)"""
    e = json.load(open(os.path.join(golden_util.HERE, "manifest_sha.json")))["cli_english5m_w1_t4"]
    data = synth.english_like(**e["generator_args"]).tobytes()
    assert hashlib.sha256(data).hexdigest() == e["input_sha256"]
    f, z, b = tmp_path / "big", tmp_path / "big.x3", tmp_path / "back"
    f.write_bytes(data)
    r = run(["-z", *e["args"], str(f)])
    assert r.returncode == 0, r.stderr.decode()
    out = z.read_bytes()
    assert len(out) == e["output_len"] and hashlib.sha256(out).hexdigest() == e["output_sha256"]
    r = run(["-d", str(z), str(b)])
    assert r.returncode == 0, r.stderr.decode()
    assert b.read_bytes() == data


def test_cli_chunked_container_round_trip(tmp_path, oracle):
    """--chunk-kib: every chunk of the X3C1 container is the stream `x3 -z` writes for that chunk alone (checked against the oracle and
    against the CLI itself), `x3 -d` recognises the container; --gpus with two handles on GPU 0; one chunk stays a raw stream."""
    import oracle_lib
    data = synth.english_like(1_000_000, seed=12).tobytes()
    f, z, b = tmp_path / "in", tmp_path / "in.x3c", tmp_path / "back"
    f.write_bytes(data)
    r = run(["-z", "-w", "64", "-t", "256", "--chunk-kib", "256", str(f), str(z)])
    assert r.returncode == 0, r.stderr.decode()
    params, chunks = container.unpack(z.read_bytes())
    assert params == dict(window_bytes=65536, max_match_count=256, factor1=4, factor2=0, nl_mode=0) and len(chunks) == 4
    cb = 256 << 10
    for i, (raw, s) in enumerate(chunks):
        part = data[i * cb:(i + 1) * cb]
        assert raw == len(part)
        if i in (0, 3):
            assert s == oracle.compress(part, oracle_lib.params(w_kib=64, t=256)), f"chunk {i}"
        p = tmp_path / f"part{i}"
        p.write_bytes(part)
        assert run(["-z", "-w", "64", "-t", "256", str(p)]).returncode == 0
        assert (tmp_path / f"part{i}.x3").read_bytes() == s, f"chunk {i} != x3 -z of that chunk"
    assert run(["-d", str(z), str(b)]).returncode == 0 and b.read_bytes() == data
    z2 = tmp_path / "two.x3c"
    r = run(["-z", "-w", "64", "-t", "256", "--chunk-kib", "256", "--gpus", "0,0", str(f), str(z2)])
    assert r.returncode == 0 and z2.read_bytes() == z.read_bytes()
    assert run(["-d", "-f", "--gpus", "0,0", str(z2), str(b)]).returncode == 0 and b.read_bytes() == data
    z4 = tmp_path / "small_batches.x3c"   # --batch-mib: the chunks coded 1 MiB at a time (bounds the workspace; default 64): same container
    r = run(["-z", "-w", "64", "-t", "256", "--chunk-kib", "64", "--batch-mib", "1", str(f), str(z4)])
    assert r.returncode == 0, r.stderr.decode()
    r = run(["-z", "-w", "64", "-t", "256", "--chunk-kib", "64", str(f), str(tmp_path / "default_batches.x3c")])
    assert r.returncode == 0 and z4.read_bytes() == (tmp_path / "default_batches.x3c").read_bytes()
    assert run(["-d", "-f", "--batch-mib", "1", str(z4), str(b)]).returncode == 0 and b.read_bytes() == data
    z3 = tmp_path / "one.x3"
    assert run(["-z", "-w", "64", "-t", "256", "--chunk-kib", "1024", str(f), str(z3)]).returncode == 0
    assert run(["-z", "-w", "64", "-t", "256", str(f)]).returncode == 0            # -> in.x3, the plain single stream
    assert z3.read_bytes()[:4] != b"X3C1" and z3.read_bytes() == (tmp_path / "in.x3").read_bytes()
    trunc = tmp_path / "trunc.x3c"
    trunc.write_bytes(z.read_bytes()[:-8])
    r = run(["-d", str(trunc), str(tmp_path / "nope")])
    assert r.returncode != 0 and b"corrupt" in r.stderr
