"""sage2_amd -- ctypes binding of libsage2ov.so, the MI355X-native SAGE2 read-overlap path (steps 1-3, and the graph simplification of step 4).

The product is the C-ABI library (include/sage2ov.h) built from sage2_amd/csrc by __graft_entry__.build()
or `make -C sage2_amd/csrc`.  This module only loads it and wraps the calls; it never computes anything
itself and has no CPU fallback: if the library is missing, or no GPU is usable, calls raise.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SAGE2OV_LIB") or os.path.join(_HERE, "libsage2ov.so")     # override only for A/B builds


class Sage2ovError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"sage2ov error {code}: {msg}")
        self.code = code


class Config(C.Structure):
    _fields_ = [("min_overlap", C.c_uint32), ("device", C.c_int32), ("rank", C.c_uint32), ("world", C.c_uint32),
                ("host_threads", C.c_uint32), ("flags", C.c_uint32)]


class ReadStats(C.Structure):
    _fields_ = [("total_reads", C.c_uint64), ("good_reads", C.c_uint64), ("unique_reads", C.c_uint64), ("total_bp", C.c_uint64),
                ("average_read_length", C.c_uint64), ("max_read_length", C.c_uint32), ("words_per_read", C.c_uint32)]


class IndexStats(C.Structure):
    _fields_ = [("slots", C.c_uint64), ("keys", C.c_uint64), ("csr_entries", C.c_uint64), ("long_buckets", C.c_uint64),
                ("hash_string_length", C.c_uint32), ("rebuilds", C.c_uint32), ("minimiser_groups", C.c_uint32), ("reserved", C.c_uint32)]


class OverlapStats(C.Structure):
    _fields_ = [("verified_overlaps", C.c_uint64), ("contained_extension", C.c_uint64), ("contained_size", C.c_uint64),
                ("left_to_explore", C.c_uint64), ("edges_inserted", C.c_uint64), ("transitive_removed", C.c_uint64),
                ("edges", C.c_uint64), ("unresolved_hits", C.c_uint64)]


class SimplifyStats(C.Structure):
    _fields_ = [("nodes_contracted", C.c_uint64), ("removed", C.c_uint64), ("loop_iterations", C.c_uint64), ("edges", C.c_uint64),
                ("reads_on_edges", C.c_uint64), ("device_ms", C.c_double)]


class Timings(C.Structure):
    _fields_ = [("index_ms", C.c_double), ("probe_ms", C.c_double), ("reciprocal_ms", C.c_double), ("reduce_ms", C.c_double),
                ("convert_ms", C.c_double), ("total_ms", C.c_double), ("probe_kernel_ms", C.c_double), ("probe_kernel_launches", C.c_uint64), ("sequential_reads", C.c_uint64),
                ("organize_ms", C.c_double), ("probe_fast_launches", C.c_uint64), ("reciprocal_cond_ms", C.c_double), ("reduce_marks_ms", C.c_double)]


class SynthParams(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("genome_len", C.c_uint64), ("n_reads", C.c_uint64), ("read_len", C.c_uint32),
                ("read_len_min", C.c_uint32), ("err_ppm", C.c_uint32), ("n_repeat_families", C.c_uint32),
                ("repeat_copies", C.c_uint32), ("repeat_len", C.c_uint32)]


EDGE_DTYPE = np.dtype([("from", "<u8"), ("to", "<u8"), ("length", "<u4"), ("length_twin", "<u4"), ("type", "u1"), ("pad", "u1", (7,))])

_lib = None


def _preload_torch_hip():
    """libsage2ov.so and PyTorch-ROCm both need `libamdhip64.so.7`; torch ships its own copy next to its other
    ROCm libraries.  Whichever copy is mapped first serves both, and torch only finds its GPUs with ITS copy, so
    when torch is installed map that one first (no torch import needed).  The C++ CLI uses /opt/rocm's."""
    if os.environ.get("SAGE2OV_NO_TORCH_HIP"):
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec and spec.origin:
            cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
            if os.path.exists(cand):
                C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except Exception:
        pass


def lib():
    """Load libsage2ov.so (fails loudly when it has not been built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` or `make -C sage2_amd/csrc`")
        _preload_torch_hip()
        L = C.CDLL(LIB_PATH)
        L.sage2ov_last_error.restype = C.c_char_p
        L.sage2ov_last_error.argtypes = [C.c_void_p]
        L.sage2ov_version.restype = C.c_char_p
        L.sage2ov_stream.restype = C.c_void_p
        L.sage2ov_stream.argtypes = [C.c_void_p]
        L.sage2ov_synth_read_len.restype = C.c_uint32
        _lib = L
    return _lib


def synth_genome(p: SynthParams) -> np.ndarray:
    g = np.zeros(p.genome_len, dtype=np.uint8)
    rc = lib().sage2ov_synth_genome(C.byref(p), C.c_void_p(g.ctypes.data))
    if rc:
        raise Sage2ovError(rc, "synth_genome")
    return g


def synth_reads_ascii(p: SynthParams, genome: np.ndarray, first=0, n=None):
    n = p.n_reads - first if n is None else n
    bases = np.zeros(n * p.read_len, dtype=np.uint8)
    off = np.zeros(n + 1, dtype=np.uint64)
    rc = lib().sage2ov_synth_reads_ascii(C.byref(p), C.c_void_p(genome.ctypes.data), C.c_uint64(first), C.c_uint64(n),
                                         C.c_void_p(bases.ctypes.data), C.c_void_p(off.ctypes.data))
    if rc:
        raise Sage2ovError(rc, "synth_reads_ascii")
    return bases[: int(off[n])], off


def synth_write_fasta(p: SynthParams, path: str):
    rc = lib().sage2ov_synth_write_fasta(C.byref(p), path.encode())
    if rc:
        raise Sage2ovError(rc, "synth_write_fasta")


class Context:
    """One overlap-detection job on one GPU (sage2ov_ctx).  Method names follow include/sage2ov.h."""

    def __init__(self, min_overlap, device=-1, rank=0, world=1, host_threads=0, flags=0):
        self._h = C.c_void_p()
        cfg = Config(min_overlap, device, rank, world, host_threads, flags)
        rc = lib().sage2ov_ctx_create(C.byref(cfg), C.byref(self._h))
        if rc:
            raise Sage2ovError(rc, lib().sage2ov_last_error(None).decode())

    def close(self):
        if self._h:
            lib().sage2ov_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc:
            raise Sage2ovError(rc, lib().sage2ov_last_error(self._h).decode())

    # ---- step 1
    def reads_add_ascii(self, bases: np.ndarray, offsets: np.ndarray):
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        self._chk(lib().sage2ov_reads_add_ascii(self._h, C.c_void_p(bases.ctypes.data), C.c_void_p(offsets.ctypes.data), C.c_uint64(len(offsets) - 1)))

    def reads_add_file(self, path1, path2=None):
        self._chk(lib().sage2ov_reads_add_file(self._h, path1.encode(), path2.encode() if path2 else None))

    def reads_add_list(self, path):
        self._chk(lib().sage2ov_reads_add_list(self._h, path.encode()))

    def reads_add_synth(self, p: SynthParams, genome: np.ndarray, first=0, n=None):
        n = p.n_reads - first if n is None else n
        self._chk(lib().sage2ov_reads_add_synth(self._h, C.byref(p), C.c_void_p(genome.ctypes.data), C.c_uint64(first), C.c_uint64(n)))

    def reads_organize(self):
        self._chk(lib().sage2ov_reads_organize(self._h))

    def reads_stats(self) -> ReadStats:
        s = ReadStats()
        self._chk(lib().sage2ov_reads_stats(self._h, C.byref(s)))
        return s

    def reads_export(self):
        s = self.reads_stats()
        n = s.unique_reads
        stride = (s.max_read_length + 3) // 4 + 1
        packed = np.zeros((n + 1, stride), dtype=np.uint8)
        length = np.zeros(n + 1, dtype=np.uint16)
        freq = np.zeros(n + 1, dtype=np.uint16)
        self._chk(lib().sage2ov_reads_export(self._h, C.c_void_p(packed.ctypes.data), C.c_uint64(stride), C.c_void_p(length.ctypes.data), C.c_void_p(freq.ctypes.data)))
        return packed, length, freq

    def reads_export_words(self):
        s = self.reads_stats()
        words = np.zeros((s.unique_reads + 1) * s.words_per_read, dtype=np.uint64)
        freq = np.zeros(s.unique_reads + 1, dtype=np.uint16)
        self._chk(lib().sage2ov_reads_export_words(self._h, C.c_void_p(words.ctypes.data), C.c_uint64(words.size), C.c_void_p(freq.ctypes.data)))
        return words, freq

    def reads_import_words(self, words, n_unique, words_per_read, max_read_length, freq, good_reads, total_bp):
        words = np.ascontiguousarray(words, dtype=np.uint64)
        freq = np.ascontiguousarray(freq, dtype=np.uint16)
        self._chk(lib().sage2ov_reads_import_words(self._h, C.c_void_p(words.ctypes.data), C.c_uint64(n_unique), C.c_uint32(words_per_read),
                                                   C.c_uint32(max_read_length), C.c_void_p(freq.ctypes.data), C.c_uint64(good_reads), C.c_uint64(total_bp)))

    def reads_save(self, path):
        self._chk(lib().sage2ov_reads_save(self._h, path.encode()))

    def reads_load(self, path):
        self._chk(lib().sage2ov_reads_load(self._h, path.encode()))

    def reads_set_totals(self, good_reads, total_bp):
        self._chk(lib().sage2ov_reads_set_totals(self._h, C.c_uint64(good_reads), C.c_uint64(total_bp)))

    # ---- step 2
    def index_build(self):
        self._chk(lib().sage2ov_index_build(self._h))

    def index_stats(self) -> IndexStats:
        s = IndexStats()
        self._chk(lib().sage2ov_index_stats_get(self._h, C.byref(s)))
        return s

    def hashtable_save(self, path):
        self._chk(lib().sage2ov_hashtable_save(self._h, path.encode()))

    def options_reload(self):
        """re-read the SAGE2OV_* environment switches (the library reads them once, when the context is created)"""
        self._chk(lib().sage2ov_options_reload(self._h))

    def index_lookup(self, v0, v1, cap=128):
        key = (C.c_uint64 * 2)(v0, v1)
        ent = (C.c_uint64 * cap)()
        cnt = C.c_uint32()
        self._chk(lib().sage2ov_index_lookup(self._h, key, ent, C.c_uint32(cap), C.byref(cnt)))
        return [int(ent[i]) for i in range(min(cnt.value, cap))], cnt.value

    # ---- step 3
    def overlap_initial(self):
        self._chk(lib().sage2ov_overlap_initial(self._h))

    def overlap_reduce(self):
        self._chk(lib().sage2ov_overlap_reduce(self._h))

    def overlap_convert(self):
        self._chk(lib().sage2ov_overlap_convert(self._h))

    def run_steps23(self):
        self._chk(lib().sage2ov_run_steps23(self._h))

    def overlap_stats(self) -> OverlapStats:
        s = OverlapStats()
        self._chk(lib().sage2ov_overlap_stats_get(self._h, C.byref(s)))
        return s

    def overlap_export_initial(self):
        n = self.reads_stats().unique_reads
        right = np.zeros(n + 1, dtype=np.uint64)
        left = np.zeros(n + 1, dtype=np.uint64)
        status = np.zeros(n + 1, dtype=np.uint8)
        conn = np.zeros(n + 1, dtype=np.uint32)
        self._chk(lib().sage2ov_overlap_export_initial(self._h, C.c_void_p(right.ctypes.data), C.c_void_p(left.ctypes.data),
                                                       C.c_void_p(status.ctypes.data), C.c_void_p(conn.ctypes.data)))
        return right, left, status, conn

    def edges(self) -> np.ndarray:
        n = C.c_uint64()
        self._chk(lib().sage2ov_edges_count(self._h, C.byref(n)))
        out = np.zeros(n.value, dtype=EDGE_DTYPE)
        if n.value:
            self._chk(lib().sage2ov_edges_export(self._h, C.c_void_p(out.ctypes.data), C.c_uint64(n.value)))
        return out

    def edges_import(self, edges: np.ndarray):
        edges = np.ascontiguousarray(edges, dtype=EDGE_DTYPE)
        self._chk(lib().sage2ov_edges_import(self._h, C.c_void_p(edges.ctypes.data), C.c_uint64(len(edges))))

    def graph_load(self, path):
        self._chk(lib().sage2ov_graph_load(self._h, path.encode()))

    def graph_save(self, path):
        self._chk(lib().sage2ov_graph_save(self._h, path.encode()))

    # ---- step 4
    def graph_simplify(self):
        self._chk(lib().sage2ov_graph_simplify(self._h))

    def simplify_stats(self) -> SimplifyStats:
        s = SimplifyStats()
        self._chk(lib().sage2ov_simplify_stats_get(self._h, C.byref(s)))
        return s

    def graph4_save(self, path):
        self._chk(lib().sage2ov_graph4_save(self._h, path.encode()))

    def timings(self) -> Timings:
        t = Timings()
        self._chk(lib().sage2ov_timings_get(self._h, C.byref(t)))
        return t

    def debug_all_hits(self):
        n = C.c_uint64()
        self._chk(lib().sage2ov_debug_all_hits(self._h, None, C.c_uint64(0), C.byref(n)))
        out = np.zeros((n.value, 5), dtype=np.uint32)
        self._chk(lib().sage2ov_debug_all_hits(self._h, C.c_void_p(out.ctypes.data), C.c_uint64(n.value), C.byref(n)))
        return out

    def timings_reset(self):
        self._chk(lib().sage2ov_timings_reset(self._h))

    def stream(self):
        return lib().sage2ov_stream(self._h)

    def debug_meminfo(self):
        """{free, total, lowest free seen by the library, workspace arena bytes} of the context's GPU"""
        o = (C.c_uint64 * 4)()
        self._chk(lib().sage2ov_debug_meminfo(self._h, o))
        return dict(free=o[0], total=o[1], lowest_free=o[2], arena=o[3])

    # ---- multi-GPU exchange points
    def shard_range(self):
        lo, hi = C.c_uint64(), C.c_uint64()
        self._chk(lib().sage2ov_shard_range(self._h, C.byref(lo), C.byref(hi)))
        return lo.value, hi.value

    def overlap_probe_shard(self):
        self._chk(lib().sage2ov_overlap_probe_shard(self._h))

    def shard_export_records(self, dev_ptr, max_reads):
        self._chk(lib().sage2ov_shard_export_records(self._h, C.c_void_p(dev_ptr), C.c_uint64(max_reads)))

    def shard_import_records(self, dev_ptr, first_id, n_reads):
        self._chk(lib().sage2ov_shard_import_records(self._h, C.c_void_p(dev_ptr), C.c_uint64(first_id), C.c_uint64(n_reads)))

    def overlap_reciprocal(self):
        self._chk(lib().sage2ov_overlap_reciprocal(self._h))

    def shard_flags_bytes(self):
        b = C.c_uint64()
        self._chk(lib().sage2ov_shard_flags_bytes(self._h, C.byref(b)))
        return b.value

    def shard_export_flags(self, dev_ptr):
        self._chk(lib().sage2ov_shard_export_flags(self._h, C.c_void_p(dev_ptr)))

    def shard_import_flags(self, dev_ptr):
        self._chk(lib().sage2ov_shard_import_flags(self._h, C.c_void_p(dev_ptr)))

    def shard_edges_count(self):
        n = C.c_uint64()
        self._chk(lib().sage2ov_shard_edges_count(self._h, C.byref(n)))
        return n.value

    def shard_edges_export(self, dev_ptr, cap):
        self._chk(lib().sage2ov_shard_edges_export(self._h, C.c_void_p(dev_ptr), C.c_uint64(cap)))

    def shard_edges_set(self, dev_ptr, n):
        self._chk(lib().sage2ov_shard_edges_set(self._h, C.c_void_p(dev_ptr), C.c_uint64(n)))

    # sharded reduce phase: this rank's survivor bucket (count, removals of its share), export, and the concatenation of all ranks' buckets back in
    def shard_survivors_count(self):
        n, rem = C.c_uint64(), C.c_uint64()
        self._chk(lib().sage2ov_shard_survivors_count(self._h, C.byref(n), C.byref(rem)))
        return n.value, rem.value

    def shard_survivors_export(self, dev_ptr, cap):
        self._chk(lib().sage2ov_shard_survivors_export(self._h, C.c_void_p(dev_ptr), C.c_uint64(cap)))

    def shard_survivors_set(self, dev_ptr, n_total, removed_total):
        self._chk(lib().sage2ov_shard_survivors_set(self._h, C.c_void_p(dev_ptr), C.c_uint64(n_total), C.c_uint64(removed_total)))
