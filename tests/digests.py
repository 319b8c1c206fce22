"""Full-size digests (tests/golden/*_digest.json): what the pinned CPU restatement (oracle/sage2_oracle.cpp) computes on the
BASELINE-size inputs, reduced to a few checksums so that the GPU path can be compared with it at sizes where the oracle itself
cannot run inside a test.  Written by oracle/make_digests.py (build container, minutes of CPU), read by the `-m gpu` full-size
tests and by bench.py (which asserts the digest of the workload it times).  Pure numpy + zlib; no oracle import here.

Digest fields (all over ids 1..N, index 0 dropped):
  n_unique, good_reads, n_ov (sum of `connections`, economyGraph.cpp:96,189,281,361), edges, contained_extension,
  contained_size, left_to_explore, edges_inserted, transitive_removed, long_buckets, keys
  conn_crc32        crc32 of connections[1..N] as little-endian u32
  right_crc32/left_crc32  crc32 of the ExtensionTable records (economyGraph.h:24-30 bit layout) as little-endian u64
  status_crc32      crc32 of the initial-pass classes (0/4/5/6; the BFS's 1/2 folded back to 0) as u8
  edges_crc32       crc32 chained over the columns from, to (u32), type (u8), length, length_twin (u32) of the canonical list
  reads_crc32       crc32 of lengths (u16) then frequencies (u16) then the packed forward strands (utils.cpp:96 bytes, ceil(L/4) per read,
                    fixed-length inputs only)
  graph3_md5/graph3_bytes  the P.graph3 file (overlapGraph.cpp:338-369)
"""
import hashlib
import json
import os
import zlib

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# BASELINE.json configs at full size (genome length and seed per config) + the secondary noisy variant
CONFIGS = {
    "c2": dict(k=40, synth=dict(seed=2, genome_len=30_000_000, n_reads=10_000_000, read_len=150)),
    "c2_noisy": dict(k=40, synth=dict(seed=2, genome_len=30_000_000, n_reads=10_000_000, read_len=150, err_ppm=1000)),
    "c3": dict(k=40, synth=dict(seed=3, genome_len=150_000_000, n_reads=50_000_000, read_len=150)),
    # read errors AND high-copy repeats: long buckets, one-sided discovery, the reduce phase's result depends on the exploration order
    "c2_repeat": dict(k=40, synth=dict(seed=2, genome_len=30_000_000, n_reads=10_000_000, read_len=150, err_ppm=1000, n_repeat_families=10, repeat_copies=300, repeat_len=400)),
    # small ones: the digest machinery itself is tested on these (CPU test: oracle vs committed digest; GPU test: device vs digest)
    "c1": dict(k=21, synth=dict(seed=1, genome_len=200_000, n_reads=100_000, read_len=100)),
    "c2_1m": dict(k=40, synth=dict(seed=2, genome_len=3_000_000, n_reads=1_000_000, read_len=150)),
}


def _crc(arr, dtype, crc=0):
    a = np.ascontiguousarray(arr, dtype=dtype)
    mv = memoryview(a).cast("B")
    step = 1 << 28
    for o in range(0, len(mv), step):
        crc = zlib.crc32(mv[o:o + step], crc)
    return crc


def status_class(status):
    s = np.asarray(status)
    return np.where((s == 1) | (s == 2), 0, s).astype(np.uint8)


def initial_digest(right, left, status, conn):
    return dict(conn_crc32=_crc(conn[1:], "<u4"), right_crc32=_crc(right[1:], "<u8"), left_crc32=_crc(left[1:], "<u8"),
                status_crc32=_crc(status_class(status[1:]), "u1"), n_ov=int(np.asarray(conn[1:], dtype=np.uint64).sum()))


def edges_digest(frm, to, typ, length, length_twin):
    c = _crc(frm, "<u4"); c = _crc(to, "<u4", c); c = _crc(typ, "u1", c); c = _crc(length, "<u4", c); c = _crc(length_twin, "<u4", c)
    return dict(edges=int(len(frm)), edges_crc32=c)


def reads_digest(packed, length, freq, read_len):
    nb = (read_len + 3) // 4
    c = _crc(length[1:], "<u2"); c = _crc(freq[1:], "<u2", c)
    p = np.asarray(packed)
    step = 1 << 22
    for o in range(1, p.shape[0], step):                       # row blocks: the slice [:, :nb] is not contiguous
        c = _crc(p[o:o + step, :nb], "u1", c)
    return dict(reads_crc32=c)


def file_digest(path):
    h = hashlib.md5(); n = 0
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 22), b""):
            h.update(blk); n += len(blk)
    return dict(graph3_md5=h.hexdigest(), graph3_bytes=n)


def path_of(name):
    return os.path.join(GOLDEN, name + "_digest.json")


def load(name):
    p = path_of(name)
    return json.load(open(p)) if os.path.exists(p) else None


def lookup(k, synth):
    """the committed digest for this exact (k, generator parameters), or None"""
    want = {kk: int(v) for kk, v in synth.items() if v}
    for name in CONFIGS:
        d = load(name)
        if d and d["k"] == k and {kk: int(v) for kk, v in d["synth"].items() if v} == want:
            return name, d
    return None, None


def gpu_digest(ctx, read_len, graph3_path=None, with_reads=True):
    """the same digest from a sage2_amd.Context after run_steps23 (through the C ABI only)"""
    d = {}
    r, l, s, c = ctx.overlap_export_initial()
    d.update(initial_digest(r, l, s, c)); del r, l, s, c
    e = ctx.edges()
    d.update(edges_digest(e["from"], e["to"], e["type"], e["length"], e["length_twin"])); del e
    if with_reads:
        p, ln, fr = ctx.reads_export()
        d.update(reads_digest(p, ln, fr, read_len)); del p, ln, fr
    st, ost = ctx.reads_stats(), ctx.overlap_stats()
    d.update(n_unique=st.unique_reads, good_reads=st.good_reads, contained_extension=ost.contained_extension, contained_size=ost.contained_size,
             left_to_explore=ost.left_to_explore, edges_inserted=ost.edges_inserted, transitive_removed=ost.transitive_removed,
             long_buckets=ctx.index_stats().long_buckets, keys=ctx.index_stats().keys)
    assert d["n_ov"] == ost.verified_overlaps, "sum of connections != the library's verified_overlaps counter"
    if graph3_path:
        ctx.graph_save(graph3_path); d.update(file_digest(graph3_path))
    return d


def compare(got, want, keys=None):
    """list of 'field: got != want' strings for every field both carry (or the given ones)"""
    bad = []
    for kk in (keys or want.keys()):
        if kk in ("k", "synth", "name", "oracle_seconds", "generated_by"):
            continue
        if kk in got and kk in want and got[kk] != want[kk]:
            bad.append(f"{kk}: {got[kk]} != {want[kk]}")
    return bad
