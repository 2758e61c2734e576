"""Chunk container for multi-stream output (new format; the reference only has the raw stream).

One chunk  -> the raw x3 code stream, byte-identical to `x3 -z` (no header at all: the CLI stays a drop-in).
Many chunks -> "X3C1" header, parameter echo, per-chunk (raw_len, comp_len) table, then the chunk streams back-to-back.
Every chunk is an independent x3 stream (own zero padding, own create() state).

The format lives in C (csrc/x3_container.c behind include/x3hip.h: the CLI and `x3h_compress_container` use the same
code); this module is the ctypes view of those entry points -- host-only functions, no GPU needed.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib

MAGIC = b"X3C1"
_LIB = None


def _lib_handle():
    global _LIB
    if _LIB is None:
        _LIB = _lib.load_library()
    return _LIB


def header(raw_lens: list[int], comp_lens: list[int], prm) -> bytes:
    """The frame in front of the chunk streams (empty for a single chunk: that one stays a raw x3 stream)."""
    assert len(raw_lens) == len(comp_lens) and len(raw_lens) >= 1
    n = len(raw_lens)
    if n == 1:
        return b""
    lib = _lib_handle()
    head = np.empty(int(lib.x3h_container_header_bytes(n)), dtype=np.uint8)
    raw = np.asarray(raw_lens, dtype=np.uint64)
    comp = np.asarray(comp_lens, dtype=np.uint64)
    rc = lib.x3h_container_write_header(head.ctypes.data, head.size, C.byref(prm), n, raw.ctypes.data, comp.ctypes.data)
    if rc != 0:
        raise ValueError(f"x3h_container_write_header: {lib.x3h_strerror(rc).decode()}")
    return head.tobytes()


def pack(streams: list[bytes], raw_lens: list[int], prm) -> bytes:
    assert len(streams) == len(raw_lens) and len(streams) >= 1
    return header(raw_lens, [len(s) for s in streams], prm) + b"".join(streams)


def unpack(blob: bytes):
    """-> (params dict or None, [(raw_len or None, stream bytes)]); ValueError on a malformed container."""
    lib = _lib_handle()
    a = np.frombuffer(bytes(blob), dtype=np.uint8)
    prm, nch, raw_total = _lib.Params(), C.c_int(0), C.c_uint64(0)
    rc = lib.x3h_container_probe(a.ctypes.data if a.size else None, a.size, C.byref(prm), C.byref(nch), C.byref(raw_total))
    if rc == _lib.NOT_A_CONTAINER:
        return None, [(None, bytes(blob))]
    if rc != 0:
        raise ValueError(f"malformed X3C1 container: {lib.x3h_strerror(rc).decode()}")
    n = nch.value
    raw = np.zeros(n, dtype=np.uint64)
    off = np.zeros(n + 1, dtype=np.uint64)
    rc = lib.x3h_container_table(a.ctypes.data, a.size, raw.ctypes.data, off.ctypes.data)
    if rc != 0:
        raise ValueError(f"malformed X3C1 container: {lib.x3h_strerror(rc).decode()}")
    params = dict(window_bytes=prm.window_bytes, max_match_count=prm.max_match_count, factor1=prm.factor1, factor2=prm.factor2,
                  nl_mode=prm.nl_mode)
    return params, [(int(raw[i]), bytes(blob[int(off[i]):int(off[i + 1])])) for i in range(n)]


def split_offsets(total: int, chunk_bytes: int):
    """Chunk boundaries for an input of `total` bytes: [0, c, 2c, ..., total]."""
    if total == 0:
        return [0, 0]
    offs = list(range(0, total, chunk_bytes)) + [total]
    return offs
