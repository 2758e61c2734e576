"""ctypes binding of oracle/libx3oracle.so -- the CPU checker.  Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ODIR = os.path.join(ROOT, "oracle")
TOK_MISS, TOK_DUP = 0x80000000, 0x40000000


class Params(C.Structure):
    _fields_ = [("window_bytes", C.c_uint32), ("max_match_count", C.c_int32), ("factor1", C.c_uint32),
                ("factor2", C.c_uint32), ("nl_mode", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("events", C.c_uint64 * 5), ("dict_elems", C.c_uint64), ("ctx0_entries", C.c_uint64), ("steps", C.c_uint64),
                ("sizes", C.c_float * 4)]


def params(w_kib=8, t=15, m=4, n=0, x=0):
    return Params(int(w_kib) * 1024, int(t), int(m), int(n), int(x))


def params_from_args(args):
    """x3 CLI arguments (['-w','64','-t','256',...]) -> Params."""
    kw = {}
    it = iter(args)
    for a in it:
        if a == "-w": kw["w_kib"] = int(next(it))
        elif a == "-t": kw["t"] = int(next(it))
        elif a == "-m": kw["m"] = int(next(it))
        elif a == "-n": kw["n"] = int(next(it))
        elif a == "-x": kw["x"] = 1
        else: raise ValueError(a)
    return params(**kw)


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        u8p, szp = C.POINTER(C.c_uint8), C.POINTER(C.c_size_t)
        lib.x3o_compress_bound.restype = C.c_size_t
        lib.x3o_compress_bound.argtypes = [C.c_size_t]
        lib.x3o_compress.argtypes = [C.POINTER(Params), C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, szp, C.POINTER(Stats)]
        lib.x3o_compress_via_m.argtypes = [C.POINTER(Params), C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, szp, C.POINTER(Stats)]
        lib.x3o_compress_trace.argtypes = [C.POINTER(Params), C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, szp, C.POINTER(Stats),
                                           C.c_void_p, C.c_void_p, C.c_size_t, szp]
        lib.x3o_decompress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, szp]
        lib.x3o_scan_m.argtypes = [C.POINTER(Params), C.c_void_p, C.c_size_t, C.c_void_p]
        lib.x3o_count.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p]
        lib.x3o_count.restype = None
        lib.x3o_ac_chain.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, szp]

    @staticmethod
    def _buf(data):
        a = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data, dtype=np.uint8)
        return a, (a.ctypes.data if a.size else None)

    def ac_chain(self, cum, freq, total):
        """ac.c:35-85 alone: -> (lo[n], hi[n]) after every symbol, from ac_init's interval"""
        cum, freq, total = (np.ascontiguousarray(x, dtype=np.uint32) for x in (cum, freq, total))
        n = cum.size
        lo, hi = np.zeros(n, np.uint32), np.zeros(n, np.uint32)
        bits = np.zeros(4 * n + 64, np.uint8)
        nb = C.c_size_t(0)
        rc = self.lib.x3o_ac_chain(cum.ctypes.data, freq.ctypes.data, total.ctypes.data, n, lo.ctypes.data, hi.ctypes.data, bits.ctypes.data, bits.size, C.byref(nb))
        assert rc == 0, rc
        return lo, hi

    def compress(self, data, prm, via_m=None, want_stats=False):
        a, ptr = self._buf(data)
        cap = self.lib.x3o_compress_bound(a.size)
        out = np.empty(cap, dtype=np.uint8)
        n_out, st = C.c_size_t(0), Stats()
        if via_m is None:
            rc = self.lib.x3o_compress(C.byref(prm), ptr, a.size, out.ctypes.data, cap, C.byref(n_out), C.byref(st))
        else:
            m = np.ascontiguousarray(via_m, dtype=np.uint8)
            rc = self.lib.x3o_compress_via_m(C.byref(prm), ptr, a.size, m.ctypes.data if m.size else None, out.ctypes.data, cap, C.byref(n_out), C.byref(st))
        assert rc == 0, f"oracle compress rc={rc}"
        res = out[:n_out.value].tobytes()
        return (res, st) if want_stats else res

    def trace(self, data, prm):
        """-> (stream bytes, tok_pos[uint32], tok_info[uint32], Stats)"""
        a, ptr = self._buf(data)
        cap = self.lib.x3o_compress_bound(a.size)
        out = np.empty(cap, dtype=np.uint8)
        tp = np.empty(a.size + 1, dtype=np.uint32)
        ti = np.empty(a.size + 1, dtype=np.uint32)
        n_out, ntok, st = C.c_size_t(0), C.c_size_t(0), Stats()
        rc = self.lib.x3o_compress_trace(C.byref(prm), ptr, a.size, out.ctypes.data, cap, C.byref(n_out), C.byref(st),
                                         tp.ctypes.data, ti.ctypes.data, tp.size, C.byref(ntok))
        assert rc == 0, f"oracle trace rc={rc}"
        return out[:n_out.value].tobytes(), tp[:ntok.value].copy(), ti[:ntok.value].copy(), st

    def decompress(self, stream, cap):
        a, ptr = self._buf(stream)
        out = np.empty(max(cap, 1), dtype=np.uint8)
        n_out = C.c_size_t(0)
        rc = self.lib.x3o_decompress(ptr, a.size, out.ctypes.data, cap, C.byref(n_out))
        return rc, out[:n_out.value].tobytes()

    def scan_m(self, data, prm):
        a, ptr = self._buf(data)
        m = np.empty(a.size, dtype=np.uint8)
        rc = self.lib.x3o_scan_m(C.byref(prm), ptr, a.size, m.ctypes.data if a.size else None)
        assert rc == 0
        return m

    def count(self, data, pos, window_bytes):
        a, _ = self._buf(data)
        padded = np.concatenate([a, np.zeros(window_bytes + 64, dtype=np.uint8)])
        cnt = np.zeros(32, dtype=np.uint32)
        self.lib.x3o_count(padded.ctypes.data, pos, window_bytes, cnt.ctypes.data)
        return cnt


def build():
    subprocess.run(["make", "-C", ODIR, "libx3oracle.so"], check=True, capture_output=True)


def load():
    so = os.path.join(ODIR, "libx3oracle.so")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(ODIR, "x3_oracle.c")):
        build()
    return Oracle(C.CDLL(so))
