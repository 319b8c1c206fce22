"""ctypes wrapper of oracle/liboracle.so -- the CPU restatement used ONLY as the checker in tests."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "liboracle.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_SO):
            subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"], check=True, stdout=subprocess.DEVNULL)
        L = C.CDLL(ORACLE_SO)
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.c_int, C.c_int]
        L.orc_counter.restype = C.c_uint64
        L.orc_counter.argtypes = [C.c_void_p, C.c_int]
        L.orc_time.restype = C.c_double
        L.orc_time.argtypes = [C.c_void_p, C.c_int]
        L.orc_get64.restype = C.c_uint64
        for f in ("orc_destroy", "orc_organize", "orc_build_index", "orc_initial", "orc_reduce", "orc_convert", "orc_run_all"):
            getattr(L, f).argtypes = [C.c_void_p]
        _lib = L
    return _lib


COUNTERS = ["N", "good_reads", "total_reads", "total_bp", "avg_len", "n_ov", "contained", "contained_size", "edges_inserted",
            "transitive_removed", "edges", "long_buckets", "M", "stride", "h", "keys"]


class Oracle:
    def __init__(self, k, threads=0):
        self.h = C.c_void_p(lib().orc_create(k, threads))

    def close(self):
        if self.h:
            lib().orc_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def add_reads_ascii(self, bases, offsets):
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        lib().orc_add_reads_ascii(self.h, C.c_void_p(bases.ctypes.data), C.c_void_p(offsets.ctypes.data), C.c_uint64(len(offsets) - 1))

    def organize(self): lib().orc_organize(self.h)
    def build_index(self): lib().orc_build_index(self.h)
    def initial(self): lib().orc_initial(self.h)
    def reduce(self): lib().orc_reduce(self.h)
    def convert(self): lib().orc_convert(self.h)
    def run_all(self): lib().orc_run_all(self.h)
    def write_reads(self, p): assert lib().orc_write_reads(self.h, p.encode()) == 0
    def write_graph3(self, p): assert lib().orc_write_graph3(self.h, p.encode()) == 0
    def counter(self, name): return int(lib().orc_counter(self.h, COUNTERS.index(name)))
    def counters(self): return {n: self.counter(n) for n in COUNTERS}
    def time(self, i): return float(lib().orc_time(self.h, i))

    def export_reads(self):
        n, stride = self.counter("N"), self.counter("stride")
        fwd = np.zeros((n + 1, stride), dtype=np.uint8)
        ln = np.zeros(n + 1, dtype=np.uint16)
        fr = np.zeros(n + 1, dtype=np.uint16)
        lib().orc_export_reads(self.h, C.c_void_p(fwd.ctypes.data), C.c_void_p(ln.ctypes.data), C.c_void_p(fr.ctypes.data))
        return fwd, ln, fr

    def export_initial(self):
        n = self.counter("N")
        right = np.zeros(n + 1, dtype=np.uint64); left = np.zeros(n + 1, dtype=np.uint64)
        status = np.zeros(n + 1, dtype=np.uint8); conn = np.zeros(n + 1, dtype=np.uint32)
        lib().orc_export_initial(self.h, C.c_void_p(right.ctypes.data), C.c_void_p(left.ctypes.data), C.c_void_p(status.ctypes.data), C.c_void_p(conn.ctypes.data))
        return right, left, status, conn

    def export_edges(self):
        n = self.counter("edges")
        e = np.zeros((n, 5), dtype=np.uint64)
        if n:
            lib().orc_export_edges(self.h, C.c_void_p(e.ctypes.data))
        return e

    def debug_hits(self, r1, cap=20000):
        out = np.zeros((cap, 3), dtype=np.int64)
        n = lib().orc_debug_hits(self.h, C.c_uint64(int(r1)), C.c_void_p(out.ctypes.data), cap)
        return out[:min(n, cap)]

    def lookup(self, v0, v1, cap=128):
        ent = (C.c_uint64 * cap)()
        n = lib().orc_lookup(self.h, C.c_uint64(v0), C.c_uint64(v1), ent, cap)
        return [int(ent[i]) for i in range(min(n, cap))], n


def pack(s: str) -> bytes:
    out = (C.c_uint8 * ((len(s) + 3) // 4))()
    lib().orc_pack(s.encode(), len(s), out)
    return bytes(out)


def get64(packed: bytes, start, length) -> int:
    buf = (C.c_uint8 * (len(packed) + 1))(*packed, 0)
    return int(lib().orc_get64(buf, start, length))
