"""Deterministic synthetic workloads for the x3 hot path (no corpus is available offline).

Two generators, both driven by a hand-written splitmix64 so that the bytes are identical on every
machine and numpy version (no dependence on numpy's distribution code):

* ``zipf_bytes``   -- BASELINE.json config 4 / SURVEY.md 8(d): i.i.d. bytes with P(byte=r) ~ 1/(r+1),
                      splitmix64 seeded 0x5EED5EED5EED5EED, the top 32 bits of each draw compared
                      against a fixed 256-entry uint32 cumulative-threshold table (``ZIPF_THRESHOLDS``,
                      integer-only, computed from exact rationals).
* ``english_like`` -- stand-in for Silesia ``dickens`` (10 192 446 bytes of English prose; the corpus is
                      not in the image and there is no network): words drawn from a Zipf-ranked
                      vocabulary (real English function words on top, pronounceable pseudo-words below),
                      sentence punctuation, capitalisation and paragraph breaks.
"""
from __future__ import annotations

import functools
from fractions import Fraction

import numpy as np

DICKENS_BYTES = 10_192_446  # size of Silesia 'dickens' 
ZIPF_SEED = 0x5EED5EED5EED5EED

_M64 = (1 << 64) - 1


def splitmix64(seed: int, n: int, offset: int = 0) -> np.ndarray:
    """n consecutive splitmix64 outputs starting at draw index `offset` (vectorised, uint64)."""
    with np.errstate(over="ignore"):
        idx = np.arange(offset + 1, offset + n + 1, dtype=np.uint64)
        z = np.uint64(seed & _M64) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _zipf_thresholds() -> np.ndarray:
    """thr[r] = floor(2^32 * sum_{j<=r} 1/(j+1) / H_256) (exact rational arithmetic), thr[255] = 2^32-1."""
    h = sum(Fraction(1, j + 1) for j in range(256))
    acc = Fraction(0)
    out = []
    for r in range(256):
        acc += Fraction(1, r + 1)
        out.append(min((acc / h * (1 << 32)).__floor__(), (1 << 32) - 1))
    out[255] = (1 << 32) - 1
    return np.array(out, dtype=np.uint64)


ZIPF_THRESHOLDS = _zipf_thresholds()


def zipf_bytes(n: int, seed: int = ZIPF_SEED, offset: int = 0) -> np.ndarray:
    """n bytes of the config-4 stream, starting at byte index `offset` of the infinite stream."""
    out = np.empty(n, dtype=np.uint8)
    step = 1 << 22
    for s in range(0, n, step):
        k = min(step, n - s)
        u = splitmix64(seed, k, offset + s) >> np.uint64(32)
        out[s:s + k] = np.searchsorted(ZIPF_THRESHOLDS, u, side="left").astype(np.uint8)
    return out


# ---------------------------------------------------------------------------------------------------
_FUNCTION_WORDS = (
    "the of and to a in that it was he his with as for had you not be her is but at on she him by which "
    "have from this all they were my are me one their so an said them we who would been will no when "
    "there if more out up into do any your what has man could other than our some very time upon about "
    "may its only now like little then can should made did us such great before must two these see know "
    "over much down after first mr good men own never most old shall day where those came come himself "
    "way work life without go make well through being long say might how am too even def again many back "
    "here think every people went same last thought away under take found hand eyes still place while "
    "just also young yet though against things get ever part nothing face right give looked house once "
    "another off put world head door night however sir room something mind moment heart always going "
    "half dear friend indeed look seemed miss knew poor took love whom better asked done told left"
).split()

_ONSETS = ["", "b", "c", "d", "f", "g", "h", "j", "k", "l", "m", "n", "p", "r", "s", "t", "v", "w",
           "st", "tr", "ch", "sh", "th", "br", "cr", "gr", "pl", "pr", "sp", "wh", "cl", "fl"]
_NUCLEI = ["a", "e", "i", "o", "u", "ea", "ou", "ee", "ai", "oo", "ie", "io"]
_CODAS = ["", "", "n", "r", "s", "t", "d", "l", "m", "ng", "nt", "st", "ck", "ll", "ss", "rd", "nd", "ly", "ed", "er"]


@functools.lru_cache(maxsize=4)
def _vocabulary(size: int, seed: int) -> list[bytes]:
    words: list[str] = list(dict.fromkeys(_FUNCTION_WORDS))
    seen = set(words)
    r = splitmix64(seed ^ 0xC0FFEE, size * 16)
    k = 0
    while len(words) < size:
        nsyl = 1 + (1 if (r[k] & 0xFF) < 120 else 0) + (1 if len(words) > 3000 and (r[k] >> 8) & 0xFF < 40 else 0)
        k += 1
        w = ""
        for _ in range(nsyl):
            v = int(r[k]); k += 1
            w += _ONSETS[v % len(_ONSETS)] + _NUCLEI[(v >> 8) % len(_NUCLEI)] + _CODAS[(v >> 16) % len(_CODAS)]
        if len(w) >= 2 and w not in seen:
            seen.add(w)
            words.append(w)
    return [w.encode("ascii") for w in words]


def english_like(n: int = DICKENS_BYTES, seed: int = 0xD1C4E25, vocab: int = 24000) -> np.ndarray:
    """n bytes of English-looking prose (ASCII), deterministic in (n, seed, vocab)."""
    voc = _vocabulary(vocab, seed)
    ranks = np.arange(1, vocab + 1, dtype=np.float64)
    # Zipf-Mandelbrot weights turned into an integer threshold table once; sampling is integer-only.
    wgt = 1.0 / (ranks + 2.7) ** 1.07
    cdf = np.cumsum(wgt)
    thr = np.floor(cdf / cdf[-1] * float(1 << 32)).astype(np.uint64)
    thr[-1] = (1 << 32) - 1

    vlen = np.array([len(w) for w in voc], dtype=np.int64)
    vblob = np.frombuffer(b"".join(voc), dtype=np.uint8)
    voff = np.concatenate(([0], np.cumsum(vlen)[:-1]))

    nwords = int(n / 4.2) + 1024  # mean token (word + separator) is > 4.2 bytes
    draws = splitmix64(seed, 2 * nwords)
    wid = np.searchsorted(thr, draws[:nwords] >> np.uint64(32), side="left")
    aux = draws[nwords:]

    # separators after each word: mostly ' ', sometimes ', ' / '. ' / '; ' / '? ' / paragraph break
    a = (aux & np.uint64(0xFFFF)).astype(np.int64)
    sep_kind = np.zeros(nwords, dtype=np.int64)          # 0: ' '
    sep_kind[a < 5200] = 1                               # ', '
    sep_kind[a < 3000] = 2                               # '. '
    sep_kind[a < 420] = 3                                # '; '
    sep_kind[a < 260] = 4                                # '? '
    sep_kind[a < 150] = 5                                # '.\n\n'
    seps = [b" ", b", ", b". ", b"; ", b"? ", b".\n\n"]
    slen = np.array([len(s) for s in seps], dtype=np.int64)
    sblob = np.frombuffer(b"".join(seps), dtype=np.uint8)
    soff = np.concatenate(([0], np.cumsum(slen)[:-1]))

    wl = vlen[wid]
    sl = slen[sep_kind]
    tok_len = wl + sl
    start = np.concatenate(([0], np.cumsum(tok_len)[:-1]))
    total = int(start[-1] + tok_len[-1])
    assert total >= n, "raise nwords"
    out = np.empty(total, dtype=np.uint8)

    # scatter word bytes, then separator bytes (vectorised ragged copy)
    def ragged_copy(dst_start, src_off, lens, blob, step=1 << 19):
        for c in range(0, len(lens), step):  # chunked: keeps the int64 temporaries cache-sized
            ln = lens[c:c + step]
            tot = int(ln.sum())
            rep = np.repeat(np.arange(len(ln)), ln)
            within = np.arange(tot) - np.repeat(np.cumsum(ln) - ln, ln)
            out[dst_start[c:c + step][rep] + within] = blob[src_off[c:c + step][rep] + within]

    ragged_copy(start, voff[wid], wl, vblob)
    ragged_copy(start + wl, soff[sep_kind], sl, sblob)

    # capitalise the first letter after a sentence end (and the very first one)
    cap = np.zeros(nwords, dtype=bool)
    cap[0] = True
    cap[1:] = np.isin(sep_kind[:-1], (2, 4, 5))
    first = start[cap]
    lower = (out[first] >= 97) & (out[first] <= 122)
    out[first[lower]] -= 32
    return out[:n].copy()


MR_BYTES = 9_970_564  # size of Silesia 'mr' 


def mr_like(n: int = MR_BYTES, seed: int = 0x4D52) -> np.ndarray:
    """n bytes standing in for Silesia ``mr`` (a 3-D MRI volume of 16-bit little-endian samples): 256 x 256
    slices, a near-zero noisy background around an elliptic 'head' whose tissue level is a sum of integer
    triangle waves plus a few bits of noise.  Integer-only (splitmix64 + shifts), deterministic in (n, seed)."""
    npix = (n + 1) // 2
    out = np.empty(2 * npix, dtype=np.uint8)
    step = 1 << 21
    for s0 in range(0, npix, step):
        k = min(step, npix - s0)
        idx = np.arange(s0, s0 + k, dtype=np.int64)
        x, y, z = idx & 255, (idx >> 8) & 255, idx >> 16
        tri = lambda v, p: np.abs(v % (2 * p) - p)
        dx, dy = x - 128, y - 120
        inside = 3 * dx * dx + 4 * dy * dy < 30000 + 400 * tri(z, 24)
        r = splitmix64(seed, k, s0)
        noise = ((r >> np.uint64(40)) & np.uint64(15)).astype(np.int64)
        bg = ((r >> np.uint64(20)) & np.uint64(3)).astype(np.int64) * (((r >> np.uint64(8)) & np.uint64(7)) == 0)
        tissue = 300 + 3 * tri(x + 3 * z, 37) + 2 * tri(2 * y + z, 53) + 5 * tri(x + y, 29) + noise
        val = np.where(inside, tissue, bg).astype(np.uint16)
        out[2 * s0:2 * (s0 + k):2] = (val & 0xFF).astype(np.uint8)
        out[2 * s0 + 1:2 * (s0 + k):2] = (val >> 8).astype(np.uint8)
    return out[:n].copy()


# sizes of the 12 Silesia files, in the corpus' alphabetical order
SILESIA = dict(dickens=10192446, mozilla=51220480, mr=9970564, nci=33553445, ooffice=6152192, osdb=10085684, reymont=6627202,
               samba=21606400, sao=7251944, webster=41458703, xray=8474240, xml=5345280)


def config3_part_spec(i: int) -> tuple[str, dict]:
    """(generator name, kwargs) of stream i of the config-3 stand-in: every third stream Zipf bytes, the others text"""
    n = list(SILESIA.values())[i]
    return ("english_like", dict(n=n, seed=1000 + i)) if i % 3 else ("zipf_bytes", dict(n=n, offset=i << 26))


def config3_part(i: int) -> np.ndarray:
    g, kw = config3_part_spec(i)
    return globals()[g](**kw)


MANY_CHUNKS_MIB = 256       # bench.py's many_chunks_batch: 1024 x 256 KiB
MANY_CHUNK_BYTES = 256 << 10
DENSE_BATCH_BYTES = 64 << 20


def many_chunks_mix(mtot: int = MANY_CHUNKS_MIB << 20) -> np.ndarray:
    """bench.py's `many_chunks_batch` input: fresh content, first half English-like text, second half Zipf(s=1) bytes"""
    q = mtot // 2
    return np.concatenate([english_like(q, seed=0xBA7C4), zipf_bytes(mtot - q, offset=1 << 33)])


def dense_batch(n: int = DENSE_BATCH_BYTES) -> np.ndarray:
    """bench.py's `many_chunks_dense_classes` input: mr-like 16-bit samples (zero background: dense n-gram classes)"""
    return mr_like(n, seed=0xBA7)


def same_bytes_chunk_range(nch: int, i: int, n: int = DICKENS_BYTES) -> tuple[int, int]:
    """(start, length) of chunk i when bench.py's `chunked_same_bytes` cuts the dickens-sized stream into `nch` chunks"""
    cb = (n + nch - 1) // nch
    return i * cb, min(cb, n - i * cb)


# pieces of the batches above whose reference streams are pinned (tests/golden/make_golden_sha.py): name -> (base generator, start, length)
def pinned_pieces() -> dict[str, tuple[str, int, int]]:
    out = {}
    cb = MANY_CHUNK_BYTES
    # text half: chunks 0..511, Zipf half: 512..1023 (the text -> Zipf boundary lies between 511 and 512)
    for c in (0, 1, 100, 255, 256, 400, 511, 512, 513, 600, 767, 768, 900, 1023):
        out[f"mcb_mix256m_chunk{c}_256k_w64_t256"] = ("many_chunks_mix", c * cb, cb)
    for c in (0, 85, 170, 255):
        out[f"mcb_dense64m_chunk{c}_256k_w64_t256"] = ("dense_batch", c * cb, cb)
    for nch in (40, 64, 128):
        for i in (0, nch - 1):
            out[f"csb_dickens_{nch}x_chunk{i}_w64_t256"] = ("english_like", *same_bytes_chunk_range(nch, i))
    return out


def workload(name: str, n: int | None = None) -> np.ndarray:
    """Named workloads used by bench.py / tests: 'dickens-like', 'zipf', 'mr-like'."""
    if name == "mr-like":
        return mr_like(MR_BYTES if n is None else n)
    if name == "dickens-like":
        return english_like(DICKENS_BYTES if n is None else n)
    if name == "zipf":
        return zipf_bytes(1 << 20 if n is None else n)
    raise ValueError(name)
