"""ctypes binding of the C ABI in include/x3hip.h.

The product library is ``csrc/libx3hip.so`` (hipcc, gfx950).  There is no CPU implementation and no fallback:
if the library is missing, or no GPU is present, construction fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_SO = os.path.join(HERE, "csrc", "libx3hip.so")

# every symbol include/x3hip.h declares (tests check that the built library exports all of them)
ABI_SYMBOLS = (
    "x3h_abi_version", "x3h_strerror", "x3h_last_hip_error", "x3h_device_count", "x3h_default_params",
    "x3h_compress_bound", "x3h_ctx_create", "x3h_ctx_destroy", "x3h_compress", "x3h_compress_chunks",
    "x3h_compress_chunks_dev", "x3h_decompress", "x3h_decompress_chunks", "x3h_decompress_chunks_dev", "x3h_scan_m", "x3h_scan_counts", "x3h_parse",
    "x3h_compress_chunks_multi", "x3h_decompress_chunks_multi", "x3h_container_header_bytes", "x3h_container_write_header",
    "x3h_container_probe", "x3h_container_table", "x3h_container_bound", "x3h_compress_container", "x3h_decompress_container",
    "x3h_coder_chain", "x3h_compress_container_rccl", "x3h_rccl_release", "x3h_ctx_set_batch_bytes", "x3h_ctx_set_estimates",
)

TOK_MISS, TOK_DUP = 0x80000000, 0x40000000


ABI_VERSION = 6
NOT_A_CONTAINER = 1  # x3h_container_probe: the bytes are one raw x3 stream


class Params(C.Structure):
    _fields_ = [("window_bytes", C.c_uint32), ("max_match_count", C.c_int32), ("factor1", C.c_uint32),
                ("factor2", C.c_uint32), ("nl_mode", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("events", C.c_uint64 * 5), ("dict_elems", C.c_uint64), ("ctx0_entries", C.c_uint64),
                ("steps", C.c_uint64), ("ms_total", C.c_double), ("ms_scan", C.c_double), ("ms_parse", C.c_double),
                ("ms_code", C.c_double), ("ms_copy", C.c_double), ("ms_features", C.c_double), ("ms_modes", C.c_double),
                ("ms_coder", C.c_double), ("ms_emit", C.c_double), ("coded_symbols", C.c_uint64), ("mode_iters", C.c_int64),
                ("chain_symbols", C.c_uint64), ("pipelined", C.c_uint64), ("est_bits", C.c_double * 4), ("coder_launches", C.c_uint64)]

    def as_dict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_ if k not in ("events", "est_bits")}
        d["events"] = list(self.events)
        d["est_bits"] = list(self.est_bits)
        return d


class X3Error(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"x3hip status {status}: {msg}")
        self.status = status


def load_library(path: str | None = None) -> C.CDLL:
    path = path or os.environ.get("X3HIP_LIBRARY") or DEFAULT_SO
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} not found: the x3 hot path only exists as HIP kernels for gfx950; build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C x3_compressor_amd/csrc`). There is no CPU fallback.")
    lib = C.CDLL(path)
    u8p = C.c_void_p
    lib.x3h_abi_version.restype = C.c_int
    if lib.x3h_abi_version() != ABI_VERSION:  # the Stats layout below mirrors include/x3hip.h of exactly this version
        raise RuntimeError(f"{path}: ABI version {lib.x3h_abi_version()}, this binding expects {ABI_VERSION} -- rebuild the library")
    lib.x3h_strerror.restype = C.c_char_p
    lib.x3h_strerror.argtypes = [C.c_int]
    lib.x3h_last_hip_error.restype = C.c_int
    lib.x3h_device_count.restype = C.c_int
    lib.x3h_default_params.argtypes = [C.POINTER(Params)]
    lib.x3h_default_params.restype = None
    lib.x3h_compress_bound.restype = C.c_size_t
    lib.x3h_compress_bound.argtypes = [C.c_size_t]
    lib.x3h_ctx_create.argtypes = [C.POINTER(C.c_void_p), C.c_int]
    lib.x3h_ctx_destroy.argtypes = [C.c_void_p]
    lib.x3h_ctx_destroy.restype = None
    lib.x3h_compress.argtypes = [C.c_void_p, C.POINTER(Params), u8p, C.c_size_t, u8p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(Stats)]
    lib.x3h_compress_chunks.argtypes = [C.c_void_p, C.POINTER(Params), u8p, C.c_void_p, C.c_int, u8p, C.c_uint64, C.c_void_p, C.POINTER(Stats)]
    lib.x3h_compress_chunks_dev.argtypes = lib.x3h_compress_chunks.argtypes
    lib.x3h_decompress.argtypes = [C.c_void_p, u8p, C.c_size_t, u8p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(Stats)]
    lib.x3h_decompress_chunks.argtypes = [C.c_void_p, u8p, C.c_void_p, C.c_int, u8p, C.c_void_p, C.c_void_p, C.POINTER(Stats)]
    lib.x3h_scan_m.argtypes = [C.c_void_p, C.POINTER(Params), u8p, C.c_size_t, u8p]
    lib.x3h_scan_counts.argtypes = [C.c_void_p, C.POINTER(Params), u8p, C.c_size_t, C.c_void_p]
    lib.x3h_parse.argtypes = [C.c_void_p, C.POINTER(Params), u8p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t,
                              C.POINTER(C.c_size_t), C.POINTER(C.c_uint64)]
    lib.x3h_coder_chain.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    ctxs = C.POINTER(C.c_void_p)
    lib.x3h_compress_chunks_multi.argtypes = [ctxs, C.c_int, C.POINTER(Params), u8p, C.c_void_p, C.c_int, u8p, C.c_uint64, C.c_void_p, C.POINTER(Stats)]
    lib.x3h_decompress_chunks_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Stats)]
    lib.x3h_decompress_chunks_multi.argtypes = [ctxs, C.c_int, u8p, C.c_void_p, C.c_int, u8p, C.c_void_p, C.c_void_p, C.POINTER(Stats)]
    lib.x3h_container_header_bytes.restype = C.c_size_t
    lib.x3h_container_header_bytes.argtypes = [C.c_int]
    lib.x3h_container_bound.restype = C.c_size_t
    lib.x3h_container_bound.argtypes = [C.c_size_t, C.c_size_t]
    lib.x3h_container_write_header.argtypes = [u8p, C.c_size_t, C.POINTER(Params), C.c_int, C.c_void_p, C.c_void_p]
    lib.x3h_container_probe.argtypes = [u8p, C.c_size_t, C.POINTER(Params), C.POINTER(C.c_int), C.POINTER(C.c_uint64)]
    lib.x3h_container_table.argtypes = [u8p, C.c_size_t, C.c_void_p, C.c_void_p]
    lib.x3h_compress_container.argtypes = [ctxs, C.c_int, C.POINTER(Params), u8p, C.c_size_t, C.c_size_t, u8p, C.c_size_t,
                                           C.POINTER(C.c_size_t), C.POINTER(Stats)]
    lib.x3h_decompress_container.argtypes = [ctxs, C.c_int, u8p, C.c_size_t, u8p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(Stats)]
    lib.x3h_compress_container_rccl.argtypes = lib.x3h_compress_container.argtypes
    lib.x3h_rccl_release.restype = None
    lib.x3h_ctx_set_batch_bytes.argtypes = [C.c_void_p, C.c_uint64]
    lib.x3h_ctx_set_estimates.argtypes = [C.c_void_p, C.c_int]
    return lib


def make_params(w_kib: int = 8, t: int = 15, m: int = 4, n: int = 0, x: int = 0) -> Params:
    """The x3 CLI knobs: -w (KiB), -t, -m, -n, -x  (x3.c:499-513)."""
    return Params(int(w_kib) * 1024, int(t), int(m), int(n), int(x))


def params_from_args(args) -> Params:
    kw = {}
    it = iter(args)
    for a in it:
        if a == "-w": kw["w_kib"] = int(next(it))
        elif a == "-t": kw["t"] = int(next(it))
        elif a == "-m": kw["m"] = int(next(it))
        elif a == "-n": kw["n"] = int(next(it))
        elif a == "-x": kw["x"] = 1
        else: raise ValueError(a)
    return make_params(**kw)


def _u8(data) -> np.ndarray:
    if isinstance(data, np.ndarray):
        return np.ascontiguousarray(data, dtype=np.uint8)
    return np.frombuffer(bytes(data), dtype=np.uint8)


class X3Context:
    """One handle per GPU (x3h_ctx): device, stream and growable HBM workspace."""

    def __init__(self, device: int = 0, library: str | None = None):
        self.lib = load_library(library)
        self._h = C.c_void_p()
        self._check(self.lib.x3h_ctx_create(C.byref(self._h), int(device)))
        self.device = device
        self.last_stats: Stats | None = None

    def close(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            self.lib.x3h_ctx_destroy(h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def set_estimates(self, on: bool = True):
        """x3h_ctx_set_estimates: compress calls also fill Stats.est_bits (the float size estimates of x3.c:43,192-193,253-266)."""
        self._check(self.lib.x3h_ctx_set_estimates(self._h, 1 if on else 0))

    def _check(self, status):
        if status != 0:
            raise X3Error(status, self.lib.x3h_strerror(status).decode() + f" (hip error {self.lib.x3h_last_hip_error()})")

    # ---- whole path ------------------------------------------------------------------------------------
    def compress(self, data, prm: Params, cap: int | None = None) -> bytes:
        a = _u8(data)
        cap = self.lib.x3h_compress_bound(a.size) if cap is None else cap
        out = np.empty(cap, dtype=np.uint8)
        n_out, st = C.c_size_t(0), Stats()
        self._check(self.lib.x3h_compress(self._h, C.byref(prm), a.ctypes.data if a.size else None, a.size,
                                          out.ctypes.data, cap, C.byref(n_out), C.byref(st)))
        self.last_stats = st
        return out[:n_out.value].tobytes()

    def compress_chunks(self, data, offsets, prm: Params, stride: int | None = None) -> list[bytes]:
        a = _u8(data)
        off = np.ascontiguousarray(offsets, dtype=np.uint64)
        nch = off.size - 1
        if stride is None:
            stride = int(self.lib.x3h_compress_bound(int(np.diff(off.astype(np.int64)).max(initial=0))))
        stride = (stride + 3) & ~3
        out = np.empty(stride * nch, dtype=np.uint8)
        lens = np.zeros(nch, dtype=np.uint64)
        st = Stats()
        self._check(self.lib.x3h_compress_chunks(self._h, C.byref(prm), a.ctypes.data if a.size else None, off.ctypes.data, nch,
                                                 out.ctypes.data, stride, lens.ctypes.data, C.byref(st)))
        self.last_stats = st
        return [out[i * stride:i * stride + int(lens[i])].tobytes() for i in range(nch)]

    def compress_chunks_dev(self, d_in: int, offsets, prm: Params, d_out: int, stride: int):
        """Device-resident call: d_in/d_out are raw device addresses (e.g. torch.Tensor.data_ptr()). -> (lens, Stats)"""
        off = np.ascontiguousarray(offsets, dtype=np.uint64)
        nch = off.size - 1
        lens = np.zeros(nch, dtype=np.uint64)
        st = Stats()
        self._check(self.lib.x3h_compress_chunks_dev(self._h, C.byref(prm), C.c_void_p(d_in), off.ctypes.data, nch,
                                                     C.c_void_p(d_out), stride, lens.ctypes.data, C.byref(st)))
        self.last_stats = st
        return lens, st

    def decompress_chunks_dev(self, d_in: int, in_offsets, d_out: int, out_offsets):
        """Device-resident decode: streams back to back at d_in (offsets multiples of 4), stream c decodes into d_out[out_offsets[c]:out_offsets[c+1]). -> (lens, Stats)"""
        ioff = np.ascontiguousarray(in_offsets, dtype=np.uint64)
        ooff = np.ascontiguousarray(out_offsets, dtype=np.uint64)
        nch = ioff.size - 1
        lens = np.zeros(nch, dtype=np.uint64)
        st = Stats()
        self._check(self.lib.x3h_decompress_chunks_dev(self._h, C.c_void_p(d_in), ioff.ctypes.data, nch, C.c_void_p(d_out), ooff.ctypes.data,
                                                       lens.ctypes.data, C.byref(st)))
        self.last_stats = st
        return lens, st

    def decompress(self, stream, cap: int) -> bytes:
        """x3h_decompress: `cap` bounds the output (the stream does not carry its length); X3Error(-3) if too small."""
        a = _u8(stream)
        out = np.empty(max(cap, 1), dtype=np.uint8)
        n_out, st = C.c_size_t(0), Stats()
        self._check(self.lib.x3h_decompress(self._h, a.ctypes.data if a.size else None, a.size, out.ctypes.data, cap, C.byref(n_out), C.byref(st)))
        self.last_stats = st
        return out[:n_out.value].tobytes()

    def decompress_chunks(self, streams: list[bytes], caps: list[int]) -> list[bytes]:
        blob = np.frombuffer(b"".join(streams), dtype=np.uint8) if sum(map(len, streams)) else np.zeros(1, np.uint8)
        ioff = np.cumsum([0] + [len(s) for s in streams]).astype(np.uint64)
        ooff = np.cumsum([0] + list(caps)).astype(np.uint64)
        out = np.empty(max(int(ooff[-1]), 1), dtype=np.uint8)
        lens = np.zeros(len(streams), dtype=np.uint64)
        st = Stats()
        self._check(self.lib.x3h_decompress_chunks(self._h, blob.ctypes.data, ioff.ctypes.data, len(streams), out.ctypes.data,
                                                   ooff.ctypes.data, lens.ctypes.data, C.byref(st)))
        self.last_stats = st
        return [out[int(ooff[i]):int(ooff[i]) + int(lens[i])].tobytes() for i in range(len(streams))]

    # ---- stage level (parity tests) ----------------------------------------------------------------------
    def scan_m(self, data, prm: Params) -> np.ndarray:
        a = _u8(data)
        m = np.empty(a.size, dtype=np.uint8)
        self._check(self.lib.x3h_scan_m(self._h, C.byref(prm), a.ctypes.data if a.size else None, a.size, m.ctypes.data if a.size else None))
        return m

    def scan_counts(self, data, prm: Params) -> np.ndarray:
        a = _u8(data)
        cnt = np.empty((a.size, 32), dtype=np.uint32)
        self._check(self.lib.x3h_scan_counts(self._h, C.byref(prm), a.ctypes.data if a.size else None, a.size, cnt.ctypes.data if a.size else None))
        return cnt

    def coder_chain(self, cum, freq, total):
        """x3h_coder_chain: -> (states[(n+7)//8, 2] = (mLow, range) before every 8th symbol, final mLow)"""
        cum, freq, total = (np.ascontiguousarray(x, dtype=np.uint32) for x in (cum, freq, total))
        n = cum.size
        states = np.zeros(((n + 7) // 8, 2), dtype=np.uint32)
        fin = C.c_uint32(0)
        self._check(self.lib.x3h_coder_chain(self._h, cum.ctypes.data, freq.ctypes.data, total.ctypes.data, n, states.ctypes.data, C.byref(fin)))
        return states, int(fin.value)

    def parse(self, data, prm: Params):
        a = _u8(data)
        tp = np.empty(a.size + 1, dtype=np.uint32)
        ti = np.empty(a.size + 1, dtype=np.uint32)
        ntok, d = C.c_size_t(0), C.c_uint64(0)
        self._check(self.lib.x3h_parse(self._h, C.byref(prm), a.ctypes.data if a.size else None, a.size, tp.ctypes.data, ti.ctypes.data,
                                       tp.size, C.byref(ntok), C.byref(d)))
        return tp[:ntok.value].copy(), ti[:ntok.value].copy(), int(d.value)


# ---- several handles at once: chunks over devices, X3C1 container (include/x3hip.h) ------------------------------------
def _handles(ctxs):
    arr = (C.c_void_p * len(ctxs))(*[c._h for c in ctxs])
    return arr, len(ctxs)


def compress_chunks_multi(ctxs: list[X3Context], data, offsets, prm: Params, stride: int | None = None) -> list[bytes]:
    """x3h_compress_chunks_multi: contiguous blocks of chunks per handle, one host thread per handle."""
    lib = ctxs[0].lib
    a = _u8(data)
    off = np.ascontiguousarray(offsets, dtype=np.uint64)
    nch = off.size - 1
    if stride is None:
        stride = int(lib.x3h_compress_bound(int(np.diff(off.astype(np.int64)).max(initial=0))))
    stride = (stride + 3) & ~3
    out = np.empty(stride * nch, dtype=np.uint8)
    lens = np.zeros(nch, dtype=np.uint64)
    st = Stats()
    arr, n = _handles(ctxs)
    ctxs[0]._check(lib.x3h_compress_chunks_multi(arr, n, C.byref(prm), a.ctypes.data if a.size else None, off.ctypes.data, nch,
                                                 out.ctypes.data, stride, lens.ctypes.data, C.byref(st)))
    ctxs[0].last_stats = st
    return [out[i * stride:i * stride + int(lens[i])].tobytes() for i in range(nch)]


def compress_container(ctxs: list[X3Context], data, prm: Params, chunk_bytes: int, rccl: bool = False) -> bytes:
    """x3h_compress_container: what `x3 -z --chunk-kib N` writes (raw stream if the input is one chunk).
    rccl=True: x3h_compress_container_rccl -- the streams stay in HBM and ONE RCCL send/receive group concatenates them on the first GPU."""
    lib = ctxs[0].lib
    a = _u8(data)
    cap = int(lib.x3h_container_bound(a.size, chunk_bytes))
    out = np.empty(cap, dtype=np.uint8)
    n_out, st = C.c_size_t(0), Stats()
    arr, n = _handles(ctxs)
    fn = lib.x3h_compress_container_rccl if rccl else lib.x3h_compress_container
    ctxs[0]._check(fn(arr, n, C.byref(prm), a.ctypes.data if a.size else None, a.size, chunk_bytes, out.ctypes.data, cap, C.byref(n_out), C.byref(st)))
    ctxs[0].last_stats = st
    return out[:n_out.value].tobytes()


def decompress_container(ctxs: list[X3Context], blob, cap: int) -> bytes:
    """x3h_decompress_container: an X3C1 container or a raw stream (what `x3 -d` reads)."""
    lib = ctxs[0].lib
    a = _u8(blob)
    out = np.empty(max(cap, 1), dtype=np.uint8)
    n_out, st = C.c_size_t(0), Stats()
    arr, n = _handles(ctxs)
    ctxs[0]._check(lib.x3h_decompress_container(arr, n, a.ctypes.data if a.size else None, a.size, out.ctypes.data, cap,
                                                C.byref(n_out), C.byref(st)))
    ctxs[0].last_stats = st
    return out[:n_out.value].tobytes()
