"""The oracle (CPU restatement) against the golden vectors produced by the real reference."""
import numpy as np
import pytest

import golden_util
import oracle_lib

CASES = sorted(golden_util.load_cases().keys())
# brute-force O(steps x window) oracle: keep the CPU suite to a couple of minutes
SLOW = {"cfg3_english16k_w256_t1024", "cfg5_english16k_w512_t4096"}


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference_stream(oracle, golden, name):
    c = golden[name]
    got = oracle.compress(c["data"], oracle_lib.params_from_args(c["args"]))
    assert got == c["expect"]


@pytest.mark.parametrize("name", CASES)
def test_oracle_decodes_reference_stream(oracle, golden, name):
    c = golden[name]
    rc, back = oracle.decompress(c["expect"], len(c["data"]) + 64)
    assert rc == 0 and back == c["data"]


@pytest.mark.parametrize("name", [n for n in CASES if n not in SLOW])
def test_closed_form_m_equals_faithful_selection(oracle, golden, name):
    """find_best_match == 1 + max{i <= m[p] : filters pass}; the GPU path relies on it."""
    c = golden[name]
    prm = oracle_lib.params_from_args(c["args"])
    m = oracle.scan_m(c["data"], prm)
    assert oracle.compress(c["data"], prm, via_m=m) == c["expect"]


@pytest.mark.parametrize("name", [n for n in CASES if n not in SLOW and n != "empty"])
def test_oracle_size_estimates_equal_reference_statistics(oracle, golden, name):
    """x3.c:43,192-193,253-266: the four float accumulators `sizes[]` (one -log2f(prob) per hit / per coded symbol of a new fragment, summed in
    single precision in coding order) -- the oracle's must reproduce the REAL reference's four statistics lines character by character."""
    c = golden[name]
    err = golden_util.reference_stderr(c["data"], c["args"])
    if err is None:
        pytest.skip("oracle/_ref/x3 not built (needs /root/reference)")
    _, st = oracle.compress(c["data"], oracle_lib.params_from_args(c["args"]), want_stats=True)
    assert golden_util.estimate_lines_from_sizes(list(st.sizes), len(c["data"])) == golden_util.estimate_lines_of(err)


def test_trace_is_consistent(oracle, golden):
    c = golden["gpl16k_default"]
    prm = oracle_lib.params_from_args(c["args"])
    stream, pos, info, st = oracle.trace(c["data"], prm)
    assert stream == c["expect"]
    assert st.steps == len(pos) and pos[0] == 0 and np.all(np.diff(pos.astype(np.int64)) > 0)
    miss = (info & oracle_lib.TOK_MISS) != 0
    lens = info[miss] & 0x3F
    assert lens.min() >= 1 and lens.max() <= 32
    inserted = miss & ((info & oracle_lib.TOK_DUP) == 0)
    assert inserted.sum() == st.dict_elems
    assert int(st.events[3]) == int(miss.sum())
    assert sum(st.events[i] for i in range(3)) == int((~miss).sum())


def test_decoder_rejects_overflow(oracle, golden):
    """The reference overruns its 64x buffer (x3.c:621) on ratio > 64:1; the restatement returns an error instead."""
    c = golden["zeros5000"]
    rc, _ = oracle.decompress(c["expect"], 100)
    assert rc == -3


def test_count_against_numpy(oracle):
    rng = np.random.default_rng(5)
    data = rng.integers(0, 3, size=600, dtype=np.uint8)
    W = 256
    padded = np.concatenate([data, np.zeros(W + 64, dtype=np.uint8)])
    for p in (0, 17, 300, 599):
        cnt = oracle.count(data, p, W)
        ref = np.zeros(32, dtype=np.uint32)
        for s in range(p + 1, p + W - 32):
            k = 0
            while k < 32 and padded[p + k] == padded[s + k]:
                ref[k] += 1
                k += 1
        assert np.array_equal(cnt, ref)


def test_corpus_hook_pins_real_files_with_the_real_reference(tmp_path, oracle):
    """tests/golden/make_golden_sha.py --corpus DIR (pin the real Silesia files the day a box has them): files named like the corpus are run through
    oracle/_ref/x3 with the arguments of the configs that name them; the manifest holds their sha256.  Here: two small stand-in files under those names; the
    manifest's streams equal the CPU oracle's for the same arguments."""
    import hashlib
    import json
    import os
    import subprocess
    import sys
    if golden_util.reference_stderr(b"abc", []) is None:
        pytest.skip("oracle/_ref/x3 not present")
    from x3_compressor_amd import synth
    files = {"dickens": synth.english_like(6000, seed=8).tobytes(), "mr": synth.mr_like(5000, seed=9).tobytes(), "notes.txt": b"not a corpus file"}
    for n, d in files.items():
        (tmp_path / n).write_bytes(d)
    out = tmp_path / "manifest.json"
    script = os.path.join(golden_util.HERE, "make_golden_sha.py")
    subprocess.run([sys.executable, script, "--corpus", str(tmp_path), "--out", str(out)], check=True, capture_output=True, timeout=600)
    man = json.load(open(out))
    assert sorted(man) == ["corpus_cfg2_dickens_w64_t256", "corpus_cfg3_dickens_w256_t1024", "corpus_cfg3_mr_w256_t1024", "corpus_cfg5_mr_w512_t4096"]
    for name, e in man.items():
        data = files[e["file"]]
        w, t = int(e["args"][e["args"].index("-w") + 1]), int(e["args"][e["args"].index("-t") + 1])
        s = oracle.compress(data, oracle_lib.params(w_kib=w, t=t))
        assert e["input_sha256"] == hashlib.sha256(data).hexdigest() and e["output_len"] == len(s) and e["output_sha256"] == hashlib.sha256(s).hexdigest(), name
