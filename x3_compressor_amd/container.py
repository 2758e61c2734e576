"""Chunk container for multi-stream output (new format; the reference only has the raw stream).

One chunk  -> the raw x3 code stream, byte-identical to `x3 -z` (no header at all: the CLI stays a drop-in).
Many chunks -> "X3C1" header, parameter echo, per-chunk (raw_len, comp_len) table, then the chunk streams back-to-back.
Every chunk is an independent x3 stream (own zero padding, own create() state), so chunk i alone decodes with `x3 -d`.
"""
from __future__ import annotations

import struct

MAGIC = b"X3C1"
_HDR = struct.Struct("<4sIIiIIiI")  # magic, version, window_bytes, max_match_count, factor1, factor2, nl_mode, nchunks
_ENT = struct.Struct("<QQ")         # raw_len, comp_len


def pack(streams: list[bytes], raw_lens: list[int], prm) -> bytes:
    assert len(streams) == len(raw_lens) and len(streams) >= 1
    if len(streams) == 1:
        return bytes(streams[0])
    head = _HDR.pack(MAGIC, 1, prm.window_bytes, prm.max_match_count, prm.factor1, prm.factor2, prm.nl_mode, len(streams))
    table = b"".join(_ENT.pack(r, len(s)) for r, s in zip(raw_lens, streams))
    return head + table + b"".join(streams)


def unpack(blob: bytes):
    """-> (params dict or None, [(raw_len or None, stream bytes)])"""
    if blob[:4] != MAGIC:
        return None, [(None, bytes(blob))]
    magic, ver, w, t, f1, f2, nl, n = _HDR.unpack_from(blob, 0)
    if ver != 1:
        raise ValueError(f"unknown container version {ver}")
    off = _HDR.size
    ents = [_ENT.unpack_from(blob, off + i * _ENT.size) for i in range(n)]
    off += n * _ENT.size
    out = []
    for raw, comp in ents:
        out.append((raw, bytes(blob[off:off + comp])))
        off += comp
    if off != len(blob):
        raise ValueError("container length mismatch")
    return dict(window_bytes=w, max_match_count=t, factor1=f1, factor2=f2, nl_mode=nl), out


def split_offsets(total: int, chunk_bytes: int):
    """Chunk boundaries for an input of `total` bytes: [0, c, 2c, ..., total]."""
    if total == 0:
        return [0, 0]
    offs = list(range(0, total, chunk_bytes)) + [total]
    return offs
