"""GPU: the HIP path (through the C ABI) against (a) the reference's own golden files, byte for byte, and
(b) the oracle's per-read intermediate results, bit for bit, on the same seeded inputs."""
import os
import numpy as np
import pytest

import fixtures as fx
import oracle_lib as ol
import sage2_amd as s2

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("minimiser_groups_on")]      # (small inputs: the groups' half of the look-up code is exercised by request, conftest.py)


def run_gpu(m, bases, off):
    ctx = s2.Context(m["k"])
    ctx.reads_add_ascii(bases, off)
    ctx.reads_organize()
    ctx.run_steps23()
    return ctx


def run_oracle(m, bases, off):
    o = ol.Oracle(m["k"], threads=8)
    o.add_reads_ascii(bases, off)
    o.organize()
    o.run_all()
    return o


def assert_equals_oracle(ctx, o):
    """canonical edge list, reduce counters, N_ov and the per-read records of a finished context against a finished oracle"""
    e, oe = ctx.edges(), o.export_edges()
    assert len(e) == len(oe) and np.array_equal(e["from"], oe[:, 0]) and np.array_equal(e["to"], oe[:, 1])
    assert np.array_equal(e["type"], oe[:, 2]) and np.array_equal(e["length"], oe[:, 3]) and np.array_equal(e["length_twin"], oe[:, 4])
    st = ctx.overlap_stats()
    assert (st.edges_inserted, st.transitive_removed, st.verified_overlaps) == (o.counter("edges_inserted"), o.counter("transitive_removed"), o.counter("n_ov"))
    assert ctx.index_stats().long_buckets == o.counter("long_buckets")
    gr, gl, gs, gc = ctx.overlap_export_initial(); orr, orl, ors, orc = o.export_initial()
    assert np.array_equal(gc, orc) and np.array_equal(gr[1:], orr[1:]) and np.array_equal(gl[1:], orl[1:])


@pytest.fixture(params=["default", "device", "device_walk", "host"])
def reduce_path(request, monkeypatch):
    """The reduce phase has two exact implementations (device: order-independent form, taken for many unresolved reads when no
    bucket is long; host: serial replay).  "device" forces the first wherever its preconditions hold, "host" the second; "device_walk" is the
    device form with the marks' short cut (kernels_reduce.inc: ra_shortcut) switched off -- every neighbour's list walked, as until round 4."""
    monkeypatch.delenv("SAGE2OV_DEVICE_REDUCE_MIN", raising=False)
    monkeypatch.delenv("SAGE2OV_HOST_REDUCE", raising=False)
    monkeypatch.delenv("SAGE2OV_RA_NO_SHORTCUT", raising=False)
    if request.param in ("device", "device_walk"):
        monkeypatch.setenv("SAGE2OV_DEVICE_REDUCE_MIN", "1")
        if request.param == "device_walk":
            monkeypatch.setenv("SAGE2OV_RA_NO_SHORTCUT", "1")
    elif request.param == "host":
        monkeypatch.setenv("SAGE2OV_HOST_REDUCE", "1")
    return request.param


@pytest.mark.parametrize("name", fx.golden_names_all())
def test_files_identical_to_reference(name, tmp_path, reduce_path):
    m = fx.golden(name)
    bases, off = fx.make_reads(m["synth"])
    ctx = run_gpu(m, bases, off)
    rp, gp = str(tmp_path / "t.reads"), str(tmp_path / "t.graph3")
    ctx.reads_save(rp)
    ctx.graph_save(gp)
    assert fx.md5_file(rp) == m["reads_md5"]
    assert fx.graph3_matches(gp, name)
    st, ref = ctx.overlap_stats(), m["counters"]
    assert st.contained_extension == ref["contained_extension"]
    assert st.contained_size == ref["contained_size"]
    assert st.left_to_explore == ref["left_to_explore"]
    assert st.edges_inserted == ref["edges_inserted"]
    assert st.transitive_removed == ref["transitive_removed"]
    assert ctx.index_stats().long_buckets == ref["long_buckets"]
    ctx.close()


@pytest.mark.parametrize("name", fx.golden_names())
def test_per_read_results_identical_to_oracle(name):
    m = fx.golden(name)
    bases, off = fx.make_reads(m["synth"])
    ctx, o = run_gpu(m, bases, off), run_oracle(m, bases, off)
    gr, gl, gs, gc = ctx.overlap_export_initial()
    orr, orl, ors, orc = o.export_initial()
    assert np.array_equal(gc, orc), "connections differ"
    assert np.array_equal(gr[1:], orr[1:]), "right extension records differ"
    assert np.array_equal(gl[1:], orl[1:]), "left extension records differ"
    # the oracle's status array has been advanced by its BFS (1/2); compare the initial-pass classes
    cls = lambda s: np.where(np.isin(s, (1, 2)), 0, s)
    assert np.array_equal(cls(gs[1:]), cls(ors[1:])), "status differs"
    assert ctx.overlap_stats().verified_overlaps == o.counter("n_ov")
    e, oe = ctx.edges(), o.export_edges()
    assert len(e) == len(oe)
    assert np.array_equal(e["from"], oe[:, 0]) and np.array_equal(e["to"], oe[:, 1]) and np.array_equal(e["type"], oe[:, 2])
    assert np.array_equal(e["length"], oe[:, 3]) and np.array_equal(e["length_twin"], oe[:, 4])
    # packed reads (utils.cpp:96 byte image), lengths and frequencies
    gp, gl_, gf = ctx.reads_export(); op, ol_, of = o.export_reads()
    w = min(gp.shape[1], op.shape[1])
    assert np.array_equal(gp[:, :w - 1], op[:, :w - 1]) and np.array_equal(gl_, ol_) and np.array_equal(gf, of)
    ctx.close(); o.close()


def test_index_buckets_match_oracle():
    """bucket contents and order for sampled keys, including long buckets (hashTable.cpp:111-123)"""
    m = fx.golden("g4_highcopy_k21")
    bases, off = fx.make_reads(m["synth"])
    ctx = s2.Context(m["k"]); ctx.reads_add_ascii(bases, off); ctx.reads_organize(); ctx.index_build()
    o = ol.Oracle(m["k"], 8); o.add_reads_ascii(bases, off); o.organize(); o.build_index()
    fwd, ln, _ = o.export_reads()
    h = m["k"]
    rng = np.random.default_rng(7)
    n_long = n_multi = 0
    for rid in rng.integers(1, len(ln), 400):
        L = int(ln[rid])
        for start in (0, L - h, int(rng.integers(0, L - h + 1))):
            v1 = ol.get64(bytes(fwd[rid]), start, h)          # h <= 32 here
            want, wn = o.lookup(0, v1)
            got, gn = ctx.index_lookup(0, v1)
            assert gn == wn and got == want[:len(got)]
            n_multi += wn > 1
    assert n_multi > 50
    assert ctx.index_stats().long_buckets == o.counter("long_buckets") == m["counters"]["long_buckets"]
    ctx.close(); o.close()


def test_roundtrip_reads_file_and_tiny_inputs(tmp_path):
    """P.reads written by us loads back (loadReadsFromFile, readLoader.cpp:289) to the same graph; tiny and
    degenerate inputs (far below the reference's own N>=12501 limit) run and agree with the oracle."""
    m = fx.golden("g3_noisy_rep_k21")
    bases, off = fx.make_reads(m["synth"])
    ctx = run_gpu(m, bases, off)
    rp = str(tmp_path / "t.reads"); ctx.reads_save(rp)
    st = ctx.reads_stats()
    c2 = s2.Context(m["k"]); c2.reads_load(rp); c2.reads_set_totals(st.good_reads, st.total_bp); c2.run_steps23()
    gp = str(tmp_path / "t2.graph3"); c2.graph_save(gp)
    assert open(gp, "rb").read() == fx.golden_graph3("g3_noisy_rep_k21")
    ctx.close(); c2.close()
    for pd, k in ((dict(seed=11, genome_len=2000, n_reads=600, read_len=80), 21),
                  (dict(seed=12, genome_len=900, n_reads=64, read_len=150), 40),
                  (dict(seed=13, genome_len=3000, n_reads=900, read_len=120, read_len_min=60, err_ppm=3000), 25)):
        b, o_ = fx.make_reads(pd)
        mm = dict(k=k)
        g, o = run_gpu(mm, b, o_), run_oracle(mm, b, o_)
        e, oe = g.edges(), o.export_edges()
        assert len(e) == len(oe) and np.array_equal(e["to"], oe[:, 1]) and np.array_equal(e["length"], oe[:, 3])
        assert g.overlap_stats().verified_overlaps == o.counter("n_ov")
        g.close(); o.close()
    # empty input
    g = s2.Context(21); g.reads_add_ascii(np.zeros(0, np.uint8), np.zeros(1, np.uint64)); g.reads_organize(); g.run_steps23()
    assert len(g.edges()) == 0
    g.close()


@pytest.mark.parametrize("pd,k", [
    (dict(seed=1002, genome_len=300000, n_reads=100000, read_len=150), 40),
    (dict(seed=1003, genome_len=160000, n_reads=80000, read_len=100, err_ppm=1500, n_repeat_families=3, repeat_copies=6, repeat_len=400), 21),
])
def test_scale_100k_matches_oracle_and_is_deterministic(pd, k, reduce_path):
    """Waves process several reads each and the table is built under real contention at this size (the golden
    fixtures are too small for either); results must still be bit-identical to the oracle, run after run."""
    bases, off = fx.make_reads(pd)
    m = dict(k=k)
    o = run_oracle(m, bases, off)
    oe = o.export_edges()
    prev = None
    for rep in range(3 if reduce_path == "default" else 1):
        g = run_gpu(m, bases, off)
        assert g.index_stats().keys == o.counter("keys")
        assert g.overlap_stats().verified_overlaps == o.counter("n_ov")
        gr, gl, gs, gc = g.overlap_export_initial(); orr, orl, ors, orc = o.export_initial()
        assert np.array_equal(gc, orc) and np.array_equal(gr[1:], orr[1:]) and np.array_equal(gl[1:], orl[1:])
        e = g.edges()
        assert len(e) == len(oe) and np.array_equal(e["from"], oe[:, 0]) and np.array_equal(e["to"], oe[:, 1])
        assert np.array_equal(e["type"], oe[:, 2]) and np.array_equal(e["length"], oe[:, 3]) and np.array_equal(e["length_twin"], oe[:, 4])
        st = g.overlap_stats()
        assert (st.edges_inserted, st.transitive_removed) == (o.counter("edges_inserted"), o.counter("transitive_removed"))
        cur = e.tobytes()
        assert prev is None or cur == prev
        prev = cur
        g.close()
    o.close()


class _Hip:
    """bare hipMalloc/hipMemcpy through ctypes (the runtime libsage2ov.so already mapped): device buffers for the
    exchange test without bringing a second framework into the test process"""
    def __init__(self):
        import ctypes as C
        self.C = C
        self.rt = C.CDLL("libamdhip64.so.7")
        self.rt.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.rt.hipFree.argtypes = [C.c_void_p]
        self.bufs = []

    def alloc(self, n):
        p = self.C.c_void_p()
        assert self.rt.hipMalloc(self.C.byref(p), max(n, 1)) == 0
        self.bufs.append(p)
        return p.value

    def to_host(self, ptr, n):
        out = np.zeros(n, dtype=np.uint8)
        if n:
            assert self.rt.hipMemcpy(out.ctypes.data, ptr, n, 2) == 0      # hipMemcpyDeviceToHost
        return out

    def to_dev(self, arr):
        arr = np.ascontiguousarray(arr, dtype=np.uint8)
        p = self.alloc(arr.size)
        if arr.size:
            assert self.rt.hipMemcpy(p, arr.ctypes.data, arr.size, 1) == 0  # hipMemcpyHostToDevice
        return p

    def free(self):
        for p in self.bufs:
            self.rt.hipFree(p)
        self.bufs = []


@pytest.mark.parametrize("name,world,device_reduce", [("g2_clean150_k40", 2, False), ("g5_mixedlen_k21", 3, False), ("g3_noisy_rep_k21", 2, False), ("g4_highcopy_k21", 2, False),
                                                      ("g3_noisy_rep_k21", 3, True), ("g4_highcopy_k21", 3, True), ("g9_repeats160k_k40", 2, True), ("g5_mixedlen_k21", 2, True)])
def test_sharded_contexts_on_one_gpu_match_reference(name, world, device_reduce, tmp_path, monkeypatch):
    if device_reduce:                                                  # the device forms of the reduce phase (symmetric / ranked), their marks sharded over the ranks
        monkeypatch.setenv("SAGE2OV_DEVICE_REDUCE_MIN", "1")
    """The multi-GPU path without a cluster (SURVEY section 4): `world` rank contexts on ONE GPU, the collectives
    replaced by explicit concatenation / element-wise max of the very buffers the C ABI exports and imports.
    Every rank must end with the reference's P.graph3."""
    from sage2_amd.shard import RECORD_BYTES, EDGE_BYTES, shard_range, max_shard      # torch-free on purpose: this process only uses the C ABI
    m = fx.golden(name)
    bases, off = fx.make_reads(m["synth"])
    hip = _Hip()
    ctxs = []
    for r in range(world):
        c = s2.Context(m["k"], device=0, rank=r, world=world)
        c.reads_add_ascii(bases, off); c.reads_organize(); c.index_build(); c.overlap_probe_shard()
        ctxs.append(c)
    n = ctxs[0].reads_stats().unique_reads
    ms = max_shard(n, world)
    sends, planes = [], []
    for c in ctxs:
        p = hip.alloc(ms * RECORD_BYTES); c.shard_export_records(p, ms); sends.append(p)
        q = hip.alloc(c.shard_flags_bytes()); c.shard_export_flags(q); planes.append(hip.to_host(q, c.shard_flags_bytes()))
    orp = hip.to_dev(np.maximum.reduce(planes))                       # MAX all-reduce == OR of the bit planes
    for c in ctxs:
        for r in range(world):
            lo, hi = shard_range(n, r, world)
            assert (lo, hi) == ctxs[r].shard_range()
            if hi > lo:
                c.shard_import_records(sends[r], lo, hi - lo)
        c.shard_import_flags(orp)
        c.overlap_reciprocal()
    parts = []
    for c in ctxs:
        ne = c.shard_edges_count()
        p = hip.alloc(max(ne, 1) * EDGE_BYTES); c.shard_edges_export(p, max(ne, 1)); parts.append(hip.to_host(p, ne * EDGE_BYTES))
    allb = np.concatenate(parts); total = allb.size // EDGE_BYTES
    dall = hip.to_dev(allb)
    # reduce phase: every rank marks its share of the unresolved reads; the survivor buckets are concatenated, the removal counters summed
    sparts, removed = [], 0
    for c in ctxs:
        c.shard_edges_set(dall if total else 0, total)
        c.overlap_reduce()
        ns, rem = c.shard_survivors_count(); removed += rem
        p = hip.alloc(max(ns, 1) * EDGE_BYTES); c.shard_survivors_export(p, max(ns, 1)); sparts.append(hip.to_host(p, ns * EDGE_BYTES))
    with pytest.raises(s2.Sage2ovError):
        ctxs[0].overlap_convert()                                      # a multi-rank context refuses to convert before the exchange
    alls = np.concatenate(sparts); stotal = alls.size // EDGE_BYTES
    dsurv = hip.to_dev(alls)
    if "transitive_removed" in m["counters"]:
        assert removed == m["counters"]["transitive_removed"]
    for r, c in enumerate(ctxs):
        c.shard_survivors_set(dsurv if stotal else 0, stotal, removed)
        c.overlap_convert()
        gp = str(tmp_path / f"r{r}.graph3"); c.graph_save(gp)
        assert fx.graph3_matches(gp, name), f"rank {r} differs from the reference"
        st = c.overlap_stats()
        assert st.contained_extension == m["counters"]["contained_extension"] and st.contained_size == m["counters"]["contained_size"]
    for c in ctxs:
        c.close()
    hip.free()


def test_cli_steps_1_to_3_write_reference_files(tmp_path):
    import subprocess, os
    cli = os.path.join(fx.ROOT, "sage2_amd", "sage2ov")
    m = fx.golden("g6_k70_150")
    fa = str(tmp_path / "x.fa")
    s2.synth_write_fasta(fx.synth_params(m["synth"]), fa)
    out = str(tmp_path / "out")
    subprocess.run([cli, "-f", fa, "-k", str(m["k"]), "-o", out, "-p", "t", "-M", "3"], check=True)
    assert fx.md5_file(os.path.join(out, "t.reads")) == m["reads_md5"]
    assert open(os.path.join(out, "t.graph3"), "rb").read() == fx.golden_graph3("g6_k70_150")
    # resume from step 2 with -i (main.cpp:63-74): loads P.reads, same graph
    subprocess.run([cli, "-f", fa, "-k", str(m["k"]), "-o", out, "-p", "u", "-i", "t", "-m", "2", "-M", "3"], check=True)
    got = open(os.path.join(out, "u.graph3"), "rb").read().split(b"\n", 3)
    want = fx.golden_graph3("g6_k70_150").split(b"\n", 3)
    assert got[3] == want[3]          # edge records identical (the 3 header lines need the FASTA totals, which P.reads lacks)
    assert os.path.exists(os.path.join(out, "t.log"))
    # `-M 2 -s` (main.cpp:63-90): steps 1-2 on the GPU, P.reads and P.hashTable -- the slot-by-slot dump of the reference's own double-hashed table
    # (hashTable.cpp:256-273) -- as the reference binary wrote them for these reads
    subprocess.run([cli, "-f", fa, "-k", str(m["k"]), "-o", out, "-p", "h", "-M", "2", "-s"], check=True)
    assert fx.md5_file(os.path.join(out, "h.reads")) == m["reads_md5"]
    assert os.path.getsize(os.path.join(out, "h.hashTable")) == m["hashtable_size"] and fx.md5_file(os.path.join(out, "h.hashTable")) == m["hashtable_md5"]


def test_device_opened_in_the_background_gives_the_same_graph_and_reports_a_bad_device_late(tmp_path):
    """SAGE2OV_FLAG_ASYNC_DEVICE (what the CLI uses: the HIP runtime starts while step 1 parses its files): create returns at once, the first call
    that needs the device joins the helper thread -- same files as the golden ones; a device that cannot be opened is reported by that call, as
    SAGE2OV_ERR_DEVICE, not swallowed and not a crash"""
    m = fx.golden("g6_k70_150")
    fa = str(tmp_path / "x.fa"); s2.synth_write_fasta(fx.synth_params(m["synth"]), fa)
    c = s2.Context(m["k"], device=0, flags=2)
    c.reads_add_file(fa); c.reads_organize(); c.run_steps23()
    gp = str(tmp_path / "t.graph3"); c.graph_save(gp); rp = str(tmp_path / "t.reads"); c.reads_save(rp)
    assert open(gp, "rb").read() == fx.golden_graph3("g6_k70_150") and fx.md5_file(rp) == m["reads_md5"]
    c.close()
    bad = s2.Context(m["k"], device=97, flags=2)           # (without the flag, create itself fails: test_abi)
    bad.reads_add_file(fa)
    with pytest.raises(s2.Sage2ovError) as ei:
        bad.reads_organize()
    assert ei.value.code == -3 and str(ei.value)
    bad.close()


@pytest.mark.parametrize("name", ["g1_clean100_k21", "g4_highcopy_k21", "g5_mixedlen_k21"])
def test_hashtable_file_from_a_device_organised_context(name, tmp_path):
    """SURVEY 8f-4 on the GPU path: reads organised ON THE DEVICE (ids, order, dedupe from k_org_*), index built, and P.hashTable written from that
    context must be the reference binary's file (md5 + size recorded by oracle/make_golden.py) -- g4 has 250 long buckets (the N+100 marker and the
    101-entry cap, hashTable.cpp:111-123,178), g5 mixed read lengths."""
    m = fx.golden(name)
    bases, off = fx.make_reads(m["synth"])
    c = s2.Context(m["k"], device=0)
    c.reads_add_ascii(bases, off); c.reads_organize(); c.index_build()
    assert c.timings().organize_ms > 0                       # step 1 really ran on the device
    p = str(tmp_path / "t.hashTable"); c.hashtable_save(p)
    assert int(open(p).readline()) == m["counters"]["hash_table_size"]
    assert os.path.getsize(p) == m["hashtable_size"] and fx.md5_file(p) == m["hashtable_md5"]
    c.close()


@pytest.mark.parametrize("pd,k", [
    (dict(seed=31, genome_len=30000, n_reads=9000, read_len=250, err_ppm=300), 40),        # 16-dword compare, 4 windows per lane
    (dict(seed=32, genome_len=30000, n_reads=9000, read_len=200, read_len_min=120, err_ppm=500), 31),   # mixed lengths in the long layout
    (dict(seed=33, genome_len=24000, n_reads=8000, read_len=150, err_ppm=0), 21),          # 130 windows: 3 windows per lane, 10-dword compare
    (dict(seed=36, genome_len=24000, n_reads=8000, read_len=150, err_ppm=2000), 22),       # the same with read errors: in-kernel state machine with 3 windows per lane
    (dict(seed=34, genome_len=40000, n_reads=6000, read_len=300, err_ppm=300), 55),        # 16-word slots (252 .. 504 bases): fast kernel with 20-dword compares, five windows per lane (round 3)
    (dict(seed=37, genome_len=60000, n_reads=9000, read_len=300, err_ppm=0), 40),          # 2 x 300 reads, error-free: the consistent path of that instantiation, 261 windows
    (dict(seed=38, genome_len=50000, n_reads=6000, read_len=320, read_len_min=255, err_ppm=1500, n_repeat_families=2, repeat_copies=6, repeat_len=600), 31),   # its state machine, mixed lengths across the 8-/16-word boundary
    (dict(seed=39, genome_len=60000, n_reads=5000, read_len=504, read_len_min=330, err_ppm=800), 64),     # 32-dword compares, eight windows per lane, the longest read of the layout
    (dict(seed=40, genome_len=50000, n_reads=4000, read_len=450, err_ppm=0), 21),          # 430 windows
    (dict(seed=35, genome_len=12000, n_reads=9000, read_len=60, err_ppm=0), 15),           # short reads, h < 16 (minimiser width = h)
])
def test_other_kernel_instantiations_match_oracle(pd, k):
    """every template instantiation of the probe kernels (slot width, compare width, windows per lane) and the
    sequential-only path, against the oracle"""
    bases, off = fx.make_reads(pd)
    m = dict(k=k)
    g, o = run_gpu(m, bases, off), run_oracle(m, bases, off)
    gr, gl, gs, gc = g.overlap_export_initial(); orr, orl, ors, orc = o.export_initial()
    assert np.array_equal(gc, orc) and np.array_equal(gr[1:], orr[1:]) and np.array_equal(gl[1:], orl[1:])
    e, oe = g.edges(), o.export_edges()
    assert len(e) == len(oe) and np.array_equal(e["from"], oe[:, 0]) and np.array_equal(e["to"], oe[:, 1])
    assert np.array_equal(e["type"], oe[:, 2]) and np.array_equal(e["length"], oe[:, 3]) and np.array_equal(e["length_twin"], oe[:, 4])
    g.close(); o.close()


def test_fastq_gz_input_and_bad_reads(tmp_path):
    """FASTQ, gzip, lower case, reads with N and reads not longer than k (readLoader.cpp:145-158, utils.cpp:144)"""
    import gzip
    m = fx.golden("g1_clean100_k21")
    bases, off = fx.make_reads(m["synth"])
    seqs = [bytes(bases[int(off[i]):int(off[i + 1])]).decode() for i in range(len(off) - 1)]
    fq = str(tmp_path / "x.fastq.gz")
    with gzip.open(fq, "wt") as f:
        for i, sq in enumerate(seqs):
            if i % 1000 == 0:
                f.write(f"@bad{i}\nACGTNNACGT{sq[:50]}\n+\n{'I' * 60}\n")            # contains N: dropped
                f.write(f"@short{i}\n{sq[:21]}\n+\n{'I' * 21}\n")                    # length == k: dropped
            s_out = sq.lower() if i % 7 == 0 else sq
            f.write(f"@r{i}\n{s_out}\n+\n{'I' * len(sq)}\n")
    c = s2.Context(m["k"]); c.reads_add_file(fq); c.reads_organize(); c.run_steps23()
    st = c.reads_stats()
    assert st.good_reads == len(seqs) and st.total_reads == len(seqs) + 2 * ((len(seqs) + 999) // 1000)
    rp, gp = str(tmp_path / "t.reads"), str(tmp_path / "t.graph3")
    c.reads_save(rp); c.graph_save(gp)
    assert fx.md5_file(rp) == m["reads_md5"] and open(gp, "rb").read() == fx.golden_graph3("g1_clean100_k21")
    c.close()


def test_bench_two_ranks_sharing_the_gpu_agree_with_one_rank():
    """bench.py's N>1 path end to end (torch.distributed.run, one process per rank, sage2_amd/dist.py): on the
    one-GPU box both ranks share cuda:0 and the collectives are staged through gloo; the canonical edge list
    (crc32 in the JSON line, asserted equal across ranks inside bench.py) must equal the single-rank one."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--reads", "200000", "--genome", "600000", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-noisy-variant"]
    env = dict(os.environ, SAGE2OV_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    one = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1"] + common, check=True, capture_output=True, text=True, timeout=240)
    r1 = json.loads(one.stdout.strip().splitlines()[-1])
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29517", os.path.join(root, "bench.py"), "--gpus", "2"] + common,
                         check=True, capture_output=True, text=True, timeout=240, env=env)
    r2 = json.loads([ln for ln in two.stdout.splitlines() if ln.startswith("{")][-1])
    assert r2["n_gpus"] == 2 and r1["n_gpus"] == 1
    for key in ("unique_reads", "verified_overlaps", "edges", "edges_crc32", "unresolved_reads"):
        assert r1["config"][key] == r2["config"][key], key
    assert r1["config"]["edges"] > 0


@pytest.mark.parametrize("pd,k", [
    (dict(seed=21, genome_len=60000, n_reads=30000, read_len=150), 40),
    (dict(seed=22, genome_len=40000, n_reads=30000, read_len=200, read_len_min=60, err_ppm=2000, n_repeat_families=2, repeat_copies=8, repeat_len=500), 21),
    (dict(seed=23, genome_len=3000, n_reads=20000, read_len=100), 21),          # 670x coverage: long runs of duplicates
])
def test_device_organizer_equals_host_organizer(pd, k, monkeypatch):
    """Step 1 on the device (canonical orientation, radix sort + tie fix-up, unique/frequency, ids) against the host
    organiser of the same library: identical read store, lengths and frequencies (ids = ranks, readLoader.cpp:215-235)."""
    bases, off = fx.make_reads(pd)
    out = {}
    for mode in ("device", "host"):
        if mode == "host":
            monkeypatch.setenv("SAGE2OV_HOST_ORGANIZE", "1")
        else:
            monkeypatch.delenv("SAGE2OV_HOST_ORGANIZE", raising=False)
        g = s2.Context(k, device=0)
        g.reads_add_ascii(bases, off); g.reads_organize()
        out[mode] = (g.reads_export(), g.reads_stats().unique_reads, g.timings().organize_ms)
        g.close()
    (dp, dl, df), dn, dms = out["device"]; (hp, hl, hf), hn, hms = out["host"]
    assert dms > 0 and hms == 0, "the device organiser must be the one that ran by default"
    assert dn == hn and np.array_equal(dl, hl) and np.array_equal(df, hf) and np.array_equal(dp, hp)
    # ... and against the pinned restatement of readLoader.cpp:179-260 (ids, lengths, frequencies, packed bytes)
    o = ol.Oracle(k, 8); o.add_reads_ascii(bases, off); o.organize()
    op, ol_, of = o.export_reads(); w = min(dp.shape[1], op.shape[1])
    assert o.counter("N") == dn and np.array_equal(dl, ol_) and np.array_equal(df, of) and np.array_equal(dp[:, :w - 1], op[:, :w - 1])
    o.close()


def test_full_size_properties_c2():
    """BASELINE configs[1] at full size (10 M x 150 bp, k=40, 30 Mb genome, error-free): too big for the oracle, so the result is
    checked through properties that do not depend on the size:
      * an error-free single-chromosome genome at 50x: after the transitive reduction the overlap graph is ONE path through all
        unique reads: N-1 edges, every read has at most one neighbour per end, exactly two reads have a free end;
      * the canonical list is strictly sorted by (from, to, type) with from < to, lengths are positive and below the read length;
      * a second run on the same context gives the same bytes (atomics-ordered build, same result)."""
    import zlib
    pd = dict(seed=2, genome_len=30_000_000, n_reads=10_000_000, read_len=150)
    p = fx.synth_params(pd)
    ctx = s2.Context(40, device=0)
    ctx.reads_add_synth(p, s2.synth_genome(p)); ctx.reads_organize(); ctx.run_steps23()
    n = ctx.reads_stats().unique_reads
    e = ctx.edges()
    assert len(e) == n - 1
    f, t, ty = e["from"].astype(np.int64), e["to"].astype(np.int64), e["type"].astype(np.int64)
    assert np.all(f < t) and np.all(f >= 1) and np.all(t <= n)
    key = (f << 34) | (t << 2) | ty
    assert np.all(key[1:] > key[:-1]), "edge list not strictly sorted by (from, to, type)"
    assert np.all(e["length"] > 0) and np.all(e["length"] < 150) and np.all(e["length_twin"] > 0) and np.all(e["length_twin"] < 150)
    # which end of each read an edge uses: type bit1 = orientation of the source, bit0 of the destination
    src_end = (ty >> 1) & 1                       # 1: leaves `from` through its right end (fwd), 0: through its left end
    dst_end = 1 - (ty & 1)                        # entering `to` forward = through its left end ... expressed as "end used": 0 left, 1 right
    use = np.zeros((n + 1, 2), dtype=np.int32)
    np.add.at(use, (f, src_end), 1); np.add.at(use, (t, dst_end), 1)
    assert use[1:].max() == 1, "a read has two neighbours on one end after the reduction"
    assert int((use[1:] == 0).sum()) == 2, "a path has exactly two free ends"
    crc = zlib.crc32(e.tobytes())
    ctx.run_steps23()
    assert zlib.crc32(ctx.edges().tobytes()) == crc
    ctx.close()


def test_high_coverage_many_candidates_match_oracle(reduce_path):
    """150x coverage: most reads have more than 128 candidates, so the fast kernel (extension records and hit lists alike) hands
    them to the sequential kernel, connection counts pass the 300 limit (status 5), and the reduce phase sees long lists."""
    pd = dict(seed=31, genome_len=20000, n_reads=30000, read_len=100, err_ppm=1000)
    bases, off = fx.make_reads(pd)
    m = dict(k=21)
    g, o = run_gpu(m, bases, off), run_oracle(m, bases, off)
    gr, gl, gs, gc = g.overlap_export_initial(); orr, orl, ors, orc = o.export_initial()
    assert np.array_equal(gc, orc) and np.array_equal(gr[1:], orr[1:]) and np.array_equal(gl[1:], orl[1:])
    assert int((gc > 128).sum()) > 1000, "the data set is meant to overflow the 128-candidate batches"
    e, oe = g.edges(), o.export_edges()
    assert len(e) == len(oe) and np.array_equal(e["from"], oe[:, 0]) and np.array_equal(e["to"], oe[:, 1])
    assert np.array_equal(e["type"], oe[:, 2]) and np.array_equal(e["length"], oe[:, 3]) and np.array_equal(e["length_twin"], oe[:, 4])
    st = g.overlap_stats()
    assert (st.edges_inserted, st.transitive_removed) == (o.counter("edges_inserted"), o.counter("transitive_removed"))
    g.close(); o.close()


@pytest.mark.parametrize("mbits", [6, 12])
def test_group_tag_collisions_are_harmless(mbits, monkeypatch):
    """Minimiser groups are found by a 24-bit tag of the minimiser: with 6-12 bits different minimisers share a group word (a group is
    then a superset, possibly oversized -> uniform table).  Results must not move; the device self-check must hold."""
    monkeypatch.setenv("SAGE2OV_TEST_MTAG_BITS", str(mbits))
    pd = dict(seed=43, genome_len=120000, n_reads=40000, read_len=150, err_ppm=500)
    bases, off = fx.make_reads(pd)
    m = dict(k=40)
    g, o = run_gpu(m, bases, off), run_oracle(m, bases, off)
    gr, gl, gs, gc = g.overlap_export_initial(); orr, orl, ors, orc = o.export_initial()
    assert np.array_equal(gc, orc) and np.array_equal(gr[1:], orr[1:]) and np.array_equal(gl[1:], orl[1:])
    e, oe = g.edges(), o.export_edges()
    assert len(e) == len(oe) and np.array_equal(e["from"], oe[:, 0]) and np.array_equal(e["to"], oe[:, 1]) and np.array_equal(e["length"], oe[:, 3])
    g.close(); o.close()


@pytest.mark.parametrize("bits", [10, 14])
def test_tag_collisions_are_harmless(bits, monkeypatch):
    """Two different keys with the same 24-bit tag on one probe chain share a bucket (about once per ten million reads).  With the tag
    cut to 10-14 bits that happens thousands of times in a small data set: merged buckets in the uniform table, their records in the
    minimiser groups of BOTH keys, ambiguous tags inside a group -- and the results must not move."""
    monkeypatch.setenv("SAGE2OV_TEST_TAG_BITS", str(bits))
    pd = dict(seed=41, genome_len=120000, n_reads=40000, read_len=150, err_ppm=500)
    bases, off = fx.make_reads(pd)
    m = dict(k=40)
    g, o = run_gpu(m, bases, off), run_oracle(m, bases, off)
    assert g.index_stats().keys < o.counter("keys"), "the shrunken tags are meant to merge buckets"
    gr, gl, gs, gc = g.overlap_export_initial(); orr, orl, ors, orc = o.export_initial()
    assert np.array_equal(gc, orc) and np.array_equal(gr[1:], orr[1:]) and np.array_equal(gl[1:], orl[1:])
    e, oe = g.edges(), o.export_edges()
    assert len(e) == len(oe) and np.array_equal(e["from"], oe[:, 0]) and np.array_equal(e["to"], oe[:, 1])
    assert np.array_equal(e["type"], oe[:, 2]) and np.array_equal(e["length"], oe[:, 3]) and np.array_equal(e["length_twin"], oe[:, 4])
    g.close(); o.close()


def test_impure_long_buckets_trigger_reseed_and_result_is_unchanged(monkeypatch, tmp_path):
    """The '>= 100 entries: hidden' verdict is the one place where a merged bucket could change the result, so long buckets are checked
    for purity on the device and an impure one reseeds the hash and rebuilds.  Never seen at 24 tag bits; with 7-bit tags the high-copy
    fixture (250 long buckets) needs several reseeds -- and P.graph3 must still be the reference's, byte for byte."""
    monkeypatch.setenv("SAGE2OV_TEST_TAG_BITS", "7")
    name = "g4_highcopy_k21"
    m = fx.golden(name)
    bases, off = fx.make_reads(m["synth"])
    ctx = run_gpu(m, bases, off)
    st = ctx.index_stats()
    assert st.rebuilds >= 1 and st.long_buckets == m["counters"]["long_buckets"]
    gp = str(tmp_path / "t.graph3"); ctx.graph_save(gp)
    assert open(gp, "rb").read() == fx.golden_graph3(name)
    ctx.close()


def test_undersized_device_buffers_are_regrown(monkeypatch):
    """The hit buffer of the device reduce is sized from an estimate and retried when a wave's chunk does not fit; the candidate list is
    regrown when the survivors do not fit.  SAGE2OV_TEST_SMALL_BUFFERS starts both far too small."""
    monkeypatch.setenv("SAGE2OV_TEST_SMALL_BUFFERS", "1")
    monkeypatch.setenv("SAGE2OV_DEVICE_REDUCE_MIN", "1")
    pd = dict(seed=1003, genome_len=160000, n_reads=80000, read_len=100, err_ppm=1500, n_repeat_families=3, repeat_copies=6, repeat_len=400)
    bases, off = fx.make_reads(pd)
    m = dict(k=21)
    g, o = run_gpu(m, bases, off), run_oracle(m, bases, off)
    e, oe = g.edges(), o.export_edges()
    assert len(e) == len(oe) and np.array_equal(e["from"], oe[:, 0]) and np.array_equal(e["to"], oe[:, 1]) and np.array_equal(e["length"], oe[:, 3])
    st = g.overlap_stats()
    assert (st.edges_inserted, st.transitive_removed) == (o.counter("edges_inserted"), o.counter("transitive_removed"))
    g.close(); o.close()


def test_palindromic_and_self_overlapping_reads_match_oracle():
    """Hand-made corner of the input space the generator never reaches: a genome that is its own reverse complement around its centre
    (reads at mirrored positions have the same canonical form; reads across the centre equal their own reverse complement), plus a
    tandem repeat with period 7 (a read overlaps its neighbours at several offsets, and its own prefix equals its own suffix window)."""
    rng = np.random.default_rng(7)
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    def rc(s): return "".join(comp[c] for c in reversed(s))
    half = "".join(rng.choice(list("ACGT"), size=700))
    genome = half + rc(half) + "".join(rng.choice(list("ACGT"), size=200)) + "ACGTTGA" * 60 + "".join(rng.choice(list("ACGT"), size=300))
    L, k, reads = 100, 21, []
    for p in range(0, len(genome) - L + 1, 3):
        s = genome[p:p + L]
        reads.append(s if (p // 3) % 2 == 0 else rc(s))
    bases = np.frombuffer("".join(reads).encode(), dtype=np.uint8).copy()
    off = np.arange(0, (len(reads) + 1) * L, L, dtype=np.uint64)
    m = dict(k=k)
    g, o = run_gpu(m, bases, off), run_oracle(m, bases, off)
    assert g.reads_stats().unique_reads == o.counter("N")
    gr, gl, gs, gc = g.overlap_export_initial(); orr, orl, ors, orc = o.export_initial()
    assert np.array_equal(gc, orc) and np.array_equal(gr[1:], orr[1:]) and np.array_equal(gl[1:], orl[1:])
    e, oe = g.edges(), o.export_edges()
    assert len(e) == len(oe) and np.array_equal(e["from"], oe[:, 0]) and np.array_equal(e["to"], oe[:, 1])
    assert np.array_equal(e["type"], oe[:, 2]) and np.array_equal(e["length"], oe[:, 3]) and np.array_equal(e["length_twin"], oe[:, 4])
    g.close(); o.close()


@pytest.mark.parametrize("k", [16, 17, 24, 32, 33, 48, 49, 63, 64, 65])
def test_key_width_boundaries_match_oracle(k):
    """Key widths around the word boundaries of the packed key (32 / 48 / 64 bases: 2, 3, 4 dwords, one or two hash rounds, h = min(k, 64))
    and around the shortest keys that still use the minimiser groups (h - 16 + 1 >= 8)."""
    pd = dict(seed=50 + k, genome_len=50000, n_reads=16000, read_len=150, err_ppm=800)
    bases, off = fx.make_reads(pd)
    m = dict(k=k)
    g, o = run_gpu(m, bases, off), run_oracle(m, bases, off)
    assert g.index_stats().keys == o.counter("keys")
    gr, gl, gs, gc = g.overlap_export_initial(); orr, orl, ors, orc = o.export_initial()
    assert np.array_equal(gc, orc) and np.array_equal(gr[1:], orr[1:]) and np.array_equal(gl[1:], orl[1:])
    e, oe = g.edges(), o.export_edges()
    assert len(e) == len(oe) and np.array_equal(e["from"], oe[:, 0]) and np.array_equal(e["to"], oe[:, 1])
    assert np.array_equal(e["type"], oe[:, 2]) and np.array_equal(e["length"], oe[:, 3]) and np.array_equal(e["length_twin"], oe[:, 4])
    g.close(); o.close()


@pytest.mark.parametrize("L", [120, 123, 124, 160, 161, 248, 250, 251, 252, 504, 505, 992, 993, 1018])
def test_read_length_layout_boundaries_match_oracle(L):
    """Longest read exactly at / one past the limits of the read-store layouts: 123 (4 words per read: 247 bits of bases above the 9-bit
    length), 160 (8 words, 10-dword compares and the in-kernel state machine), 251 (8 words, 16-dword compares; bases and length share the
    last dword), 252-504 (16 words: sequential kernel), 505-1018 (32 words, 11-bit length field; 993+: bases and length share the last word).
    Mixed lengths below it.  The packed reads are compared too."""
    pd = dict(seed=70 + L, genome_len=40000, n_reads=12000 if L < 500 else 4000, read_len=L, read_len_min=L - 40, err_ppm=1000)
    bases, off = fx.make_reads(pd)
    m = dict(k=31)
    g, o = run_gpu(m, bases, off), run_oracle(m, bases, off)
    gr, gl, gs, gc = g.overlap_export_initial(); orr, orl, ors, orc = o.export_initial()
    assert np.array_equal(gc, orc) and np.array_equal(gr[1:], orr[1:]) and np.array_equal(gl[1:], orl[1:])
    e, oe = g.edges(), o.export_edges()
    assert len(e) == len(oe) and np.array_equal(e["from"], oe[:, 0]) and np.array_equal(e["to"], oe[:, 1])
    assert np.array_equal(e["type"], oe[:, 2]) and np.array_equal(e["length"], oe[:, 3]) and np.array_equal(e["length_twin"], oe[:, 4])
    gp, gl_, gf = g.reads_export(); op, ol_, of = o.export_reads()
    w = min(gp.shape[1], op.shape[1])
    assert np.array_equal(gl_, ol_) and np.array_equal(gf, of) and np.array_equal(gp[:, :w - 1], op[:, :w - 1])
    g.close(); o.close()


def test_device_organizer_long_runs_of_equal_prefixes(monkeypatch):
    """Thousands of distinct reads that share their first 40 bases (adapter dimers, amplicons), plus heavy exact duplication: the per-run
    insertion sort of the device organiser would be quadratic, so it must switch to the all-words radix sort -- same ids as the host."""
    rng = np.random.default_rng(11)
    prefix = "".join(rng.choice(list("ACGT"), size=40))
    reads = [prefix + "".join(rng.choice(list("ACGT"), size=60)) for _ in range(6000)]
    reads += [reads[0]] * 3000 + ["".join(rng.choice(list("ACGT"), size=100)) for _ in range(8000)]
    bases = np.frombuffer("".join(reads).encode(), dtype=np.uint8).copy()
    off = np.arange(0, (len(reads) + 1) * 100, 100, dtype=np.uint64)
    out = {}
    for mode in ("device", "host"):
        if mode == "host":
            monkeypatch.setenv("SAGE2OV_HOST_ORGANIZE", "1")
        else:
            monkeypatch.delenv("SAGE2OV_HOST_ORGANIZE", raising=False)
        g = s2.Context(21, device=0); g.reads_add_ascii(bases, off); g.reads_organize()
        out[mode] = (g.reads_export(), g.reads_stats().unique_reads); g.close()
    (dp, dl, df), dn = out["device"]; (hp, hl, hf), hn = out["host"]
    assert dn == hn and np.array_equal(dl, hl) and np.array_equal(df, hf) and np.array_equal(dp, hp)
    assert int(df.max()) >= 3001
    o = ol.Oracle(21, 8); o.add_reads_ascii(bases, off); o.organize()
    op, ol_, of = o.export_reads(); w = min(dp.shape[1], op.shape[1])
    assert o.counter("N") == dn and np.array_equal(dl, ol_) and np.array_equal(df, of) and np.array_equal(dp[:, :w - 1], op[:, :w - 1])
    o.close()


def test_device_organizer_full_sort_path_equals_default(monkeypatch):
    """the all-words radix sort (normally only taken for long runs) on ordinary mixed-length data"""
    pd = dict(seed=22, genome_len=40000, n_reads=30000, read_len=200, read_len_min=60, err_ppm=2000)
    bases, off = fx.make_reads(pd)
    out = {}
    for mode in ("default", "full"):
        if mode == "full":
            monkeypatch.setenv("SAGE2OV_TEST_FULL_SORT", "1")
        g = s2.Context(21, device=0); g.reads_add_ascii(bases, off); g.reads_organize()
        out[mode] = g.reads_export(); g.close()
    for a, b in zip(out["default"], out["full"]):
        assert np.array_equal(a, b)


def test_long_bucket_reduce_ranked_device_path_equals_serial_replay(monkeypatch):
    """high-copy repeats (long buckets: one-sided discovery) with read errors, 300 k reads: the default path (exploration order on the
    host, lists / marks / removals on the device) against the serial replay of the whole phase on the host -- same edges, same counters"""
    pd = dict(seed=31, genome_len=900000, n_reads=300000, read_len=150, err_ppm=1500, n_repeat_families=5, repeat_copies=250, repeat_len=350)
    bases, off = fx.make_reads(pd)
    res = {}
    for mode in ("default", "host"):
        if mode == "host":
            monkeypatch.setenv("SAGE2OV_HOST_REDUCE", "1")
        else:
            monkeypatch.delenv("SAGE2OV_HOST_REDUCE", raising=False)
        ctx = s2.Context(40); ctx.reads_add_ascii(bases, off); ctx.reads_organize(); ctx.run_steps23()
        st = ctx.overlap_stats(); res[mode] = (ctx.edges().tobytes(), st.edges_inserted, st.transitive_removed, st.left_to_explore, ctx.index_stats().long_buckets)
        ctx.close()
    assert res["default"][4] > 0 and res["default"][3] > 100000
    assert res["default"][1:] == res["host"][1:]
    assert res["default"][0] == res["host"][0]
    # ... and both against the pinned restatement of the reference (not only against each other)
    monkeypatch.delenv("SAGE2OV_HOST_REDUCE", raising=False)
    ctx = s2.Context(40); ctx.reads_add_ascii(bases, off); ctx.reads_organize(); ctx.run_steps23()
    o = run_oracle(dict(k=40), bases, off)
    assert_equals_oracle(ctx, o)
    ctx.close(); o.close()


def test_long_bucket_reduce_with_oversized_lists_equals_serial_replay(monkeypatch):
    """a genome one quarter repeats: a few hundred unresolved reads next to the repeats are seen by thousands of others, so their potential
    and final lists exceed what the LDS kernels hold (host-sorted potential lists, `k_ra_mark_big` out of global scratch) -- same edges
    and counters as the serial replay"""
    pd = dict(seed=3, genome_len=1500000, n_reads=500000, read_len=150, err_ppm=1000, n_repeat_families=12, repeat_copies=300, repeat_len=400)
    bases, off = fx.make_reads(pd)
    res = {}
    for mode in ("default", "host"):
        if mode == "host":
            monkeypatch.setenv("SAGE2OV_HOST_REDUCE", "1")
        else:
            monkeypatch.delenv("SAGE2OV_HOST_REDUCE", raising=False)
        ctx = s2.Context(40); ctx.reads_add_ascii(bases, off); ctx.reads_organize(); ctx.run_steps23()
        st = ctx.overlap_stats(); res[mode] = (ctx.edges().tobytes(), st.edges_inserted, st.transitive_removed, st.left_to_explore, ctx.index_stats().long_buckets)
        ctx.close()
    assert res["default"][4] > 500 and res["default"][3] > 100000
    assert res["default"][1:] == res["host"][1:]
    assert res["default"][0] == res["host"][0]
    monkeypatch.delenv("SAGE2OV_HOST_REDUCE", raising=False)
    ctx = s2.Context(40); ctx.reads_add_ascii(bases, off); ctx.reads_organize(); ctx.run_steps23()
    o = run_oracle(dict(k=40), bases, off)
    assert_equals_oracle(ctx, o)
    ctx.close(); o.close()


@pytest.mark.parametrize("pd,k", [
    (dict(seed=81, genome_len=400000, n_reads=150000, read_len=150, read_len_min=80, err_ppm=2000, n_repeat_families=6, repeat_copies=200, repeat_len=300), 31),   # mixed lengths: containment + long buckets
    (dict(seed=82, genome_len=300000, n_reads=150000, read_len=100, err_ppm=500, n_repeat_families=3, repeat_copies=400, repeat_len=120), 21),                    # short repeats inside reads, k=21
    (dict(seed=83, genome_len=700000, n_reads=160000, read_len=250, err_ppm=1500, n_repeat_families=2, repeat_copies=600, repeat_len=300), 55),                   # 250-bp reads, 16-dword layout
])
def test_long_bucket_reduce_more_shapes_equal_serial_replay(pd, k, monkeypatch):
    """the ranked device reduce against the serial replay on other shapes of data: mixed read lengths (containment marks next to
    long buckets), repeats shorter than a read, the long layout"""
    bases, off = fx.make_reads(pd)
    res = {}
    for mode in ("default", "host"):
        if mode == "host":
            monkeypatch.setenv("SAGE2OV_HOST_REDUCE", "1")
        else:
            monkeypatch.delenv("SAGE2OV_HOST_REDUCE", raising=False)
            monkeypatch.setenv("SAGE2OV_DEVICE_REDUCE_MIN", "1")
        ctx = s2.Context(k); ctx.reads_add_ascii(bases, off); ctx.reads_organize(); ctx.run_steps23()
        st = ctx.overlap_stats(); res[mode] = (ctx.edges().tobytes(), st.edges_inserted, st.transitive_removed, st.left_to_explore, ctx.index_stats().long_buckets)
        ctx.close()
        monkeypatch.delenv("SAGE2OV_DEVICE_REDUCE_MIN", raising=False)
    assert res["default"][4] > 0 and res["default"][3] > 1000
    assert res["default"][1:] == res["host"][1:]
    assert res["default"][0] == res["host"][0]
    monkeypatch.setenv("SAGE2OV_DEVICE_REDUCE_MIN", "1")
    ctx = s2.Context(k); ctx.reads_add_ascii(bases, off); ctx.reads_organize(); ctx.run_steps23()
    o = run_oracle(dict(k=k), bases, off)
    assert_equals_oracle(ctx, o)
    ctx.close(); o.close()


def test_long_bucket_reduce_in_many_slices(monkeypatch):
    """the potential lists are built for a slice of the unresolved reads at a time (2^30 entries per slice; 50 M noisy reads need five):
    here with slices of 200 k entries, i.e. dozens of them, against the serial replay"""
    pd = dict(seed=3, genome_len=1500000, n_reads=500000, read_len=150, err_ppm=1000, n_repeat_families=12, repeat_copies=300, repeat_len=400)
    bases, off = fx.make_reads(pd)
    res = {}
    for mode in ("sliced", "host"):
        if mode == "host":
            monkeypatch.setenv("SAGE2OV_HOST_REDUCE", "1")
        else:
            monkeypatch.delenv("SAGE2OV_HOST_REDUCE", raising=False); monkeypatch.setenv("SAGE2OV_TEST_RANK_SLICE", "200000")
        ctx = s2.Context(40); ctx.reads_add_ascii(bases, off); ctx.reads_organize(); ctx.run_steps23()
        st = ctx.overlap_stats(); res[mode] = (ctx.edges().tobytes(), st.edges_inserted, st.transitive_removed, st.left_to_explore)
        ctx.close(); monkeypatch.delenv("SAGE2OV_TEST_RANK_SLICE", raising=False)
    assert res["sliced"][1:] == res["host"][1:] and res["sliced"][0] == res["host"][0]


def test_device_filter_and_pack_of_ascii_input(monkeypatch):
    """sage2ov_reads_add_ascii on a GPU context hands the RAW bases to the device: isGoodRead (utils.cpp:144-166: longer than minOverlap, only
    ACGTacgt, lower case accepted), charsToBytes (utils.cpp:96-119) and the canonical orientation (readLoader.cpp:195) run there.  Lower-case
    reads, reads with N / other characters, reads not longer than k and mixed lengths, in several batches -- counters, ids, lengths, frequencies and
    packed bytes against the host path of the same library (SAGE2OV_HOST_PACK) and against the oracle."""
    pd = dict(seed=77, genome_len=30000, n_reads=14000, read_len=120, read_len_min=30, err_ppm=1000)
    bases, off = fx.make_reads(pd)
    seqs = [bytes(bases[int(off[i]):int(off[i + 1])]) for i in range(len(off) - 1)]
    rng = np.random.default_rng(5)
    for i in range(0, len(seqs), 7):
        seqs[i] = seqs[i].lower()                                            # accepted, upper-cased
    for i in range(3, len(seqs), 97):
        s = bytearray(seqs[i]); s[int(rng.integers(0, len(s)))] = ord("N"); seqs[i] = bytes(s)      # dropped
    for i in range(5, len(seqs), 211):
        s = bytearray(seqs[i]); s[-1] = ord("x"); seqs[i] = bytes(s)                               # dropped (bad last character)
    flat = np.frombuffer(b"".join(seqs), dtype=np.uint8).copy()
    o2 = np.zeros(len(seqs) + 1, dtype=np.uint64); o2[1:] = np.cumsum([len(s) for s in seqs])
    k = 35                                                                    # reads of 30..35 bases are "small"
    res = {}
    for mode in ("device", "host"):
        if mode == "host":
            monkeypatch.setenv("SAGE2OV_HOST_PACK", "1")
        else:
            monkeypatch.delenv("SAGE2OV_HOST_PACK", raising=False)
        c = s2.Context(k, device=0)
        third = len(seqs) // 3                                                # three batches: offsets of later batches do not start at 0
        for a, b in ((0, third), (third, 2 * third), (2 * third, len(seqs))):
            c.reads_add_ascii(flat, o2[a:b + 1])
        c.reads_organize(); st = c.reads_stats()
        res[mode] = (c.reads_export(), (st.total_reads, st.good_reads, st.unique_reads, st.total_bp, st.max_read_length, st.words_per_read), c)
    (dp, dl, df), dst, dc = res["device"]; (hp, hl, hf), hst, hc = res["host"]
    assert dst == hst and dst[0] == len(seqs) and dst[1] < len(seqs)
    assert np.array_equal(dl, hl) and np.array_equal(df, hf) and np.array_equal(dp, hp)
    o = ol.Oracle(k, 8); o.add_reads_ascii(flat, o2); o.organize()
    op, ol_, of = o.export_reads(); w = min(dp.shape[1], op.shape[1])
    assert (o.counter("total_reads"), o.counter("good_reads"), o.counter("N"), o.counter("total_bp")) == dst[:4]
    assert np.array_equal(dl, ol_) and np.array_equal(df, of) and np.array_equal(dp[:, :w - 1], op[:, :w - 1])
    dc.run_steps23(); o.run_all(); assert_equals_oracle(dc, o)
    dc.close(); hc.close(); o.close()


def test_index_heavy_windows_match_oracle():
    """One key in 70 000 reads (all start with the same 25 bases) and one in 5 000: the table windows those keys land in hold far more tuples than a
    workgroup keeps in registers (3 072) resp. than the LDS-staged bucket order can index (65 535), so the build takes its global-scratch paths
    (k_ix_window: surplus tuples through `wh`, CSR segment written and ordered in global memory by id) and the purity check walks a 70 000-entry
    bucket.  Results against the oracle: bucket contents in order, long-bucket count, extension records, edges."""
    rng = np.random.default_rng(123)
    L, k = 100, 25
    def rnd(n): return "".join(rng.choice(list("ACGT"), size=n))
    p1, p2 = rnd(25), rnd(25)
    reads = [p1 + rnd(L - 25) for _ in range(70000)] + [p2 + rnd(L - 25) for _ in range(5000)]
    genome = rnd(60000)
    reads += [genome[s:s + L] for s in rng.integers(0, len(genome) - L, size=30000)]
    bases = np.frombuffer("".join(reads).encode(), dtype=np.uint8).copy()
    off = np.arange(0, (len(reads) + 1) * L, L, dtype=np.uint64)
    m = dict(k=k)
    g, o = run_gpu(m, bases, off), run_oracle(m, bases, off)
    assert g.reads_stats().unique_reads == o.counter("N")
    assert g.index_stats().long_buckets == o.counter("long_buckets") >= 2
    fwd, ln, _ = o.export_reads()
    for rid in rng.integers(1, len(ln), 300):                                   # sampled keys, incl. suffix keys of the heavy reads (short buckets in heavy windows)
        for start in (0, int(ln[rid]) - k):
            want, wn = o.lookup(0, ol.get64(bytes(fwd[rid]), start, k)); got, gn = g.index_lookup(0, ol.get64(bytes(fwd[rid]), start, k))
            assert gn == wn and got == want[:len(got)]
    assert_equals_oracle(g, o)
    g.close(); o.close()


def test_table_beyond_2_to_32_slots(monkeypatch):
    """BASELINE configs[4] (1.2 G reads, about 1.0 G unique) needs 8 N = 8.2 G slots: slot indices are 64-bit everywhere (pair indices and window ids
    stay below 2^32 up to 2^33 slots).  Exercised here by forcing a table of 2^32 + 2^20 slots (34 GB of HBM, written once by the window kernel)
    under a small read set: index contents, extension records and edges against the oracle."""
    monkeypatch.setenv("SAGE2OV_TEST_TABLE_SLOTS", str((1 << 32) + (1 << 20)))
    pd = dict(seed=91, genome_len=60000, n_reads=30000, read_len=150, err_ppm=500)
    bases, off = fx.make_reads(pd)
    m = dict(k=40)
    g, o = run_gpu(m, bases, off), run_oracle(m, bases, off)
    assert g.index_stats().slots >= (1 << 32) + (1 << 20) and g.index_stats().keys == o.counter("keys")
    fwd, ln, _ = o.export_reads()
    rng = np.random.default_rng(3)
    for rid in rng.integers(1, len(ln), 200):
        for start in (0, int(ln[rid]) - 40):
            b = bytes(fwd[rid]); v0, v1 = ol.get64(b, start, 8), ol.get64(b, start + 8, 32)
            want, wn = o.lookup(v0, v1); got, gn = g.index_lookup(v0, v1)
            assert gn == wn and got == want[:len(got)]
    assert_equals_oracle(g, o)
    g.close(); o.close()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["sample", "tail0", "tail1", "tail2", "tail2_own_hit_lists"])
@pytest.mark.parametrize("pd,k", [
    (dict(seed=71, genome_len=60000, n_reads=20000, read_len=150, err_ppm=0), 40),                          # clean: the sample keeps the kernel without the state machine
    (dict(seed=72, genome_len=60000, n_reads=20000, read_len=150, err_ppm=1500), 40),                       # noisy: the sample switches to the kernel with it
    (dict(seed=73, genome_len=50000, n_reads=16000, read_len=150, read_len_min=90, err_ppm=800), 31),        # mixed lengths (containments), errors
    (dict(seed=74, genome_len=30000, n_reads=12000, read_len=100, err_ppm=1000, n_repeat_families=4, repeat_len=400, repeat_copies=6), 25),   # repeats: reads past 128 candidates end in the sequential kernel
    (dict(seed=75, genome_len=40000, n_reads=9000, read_len=250, err_ppm=1000), 45),                        # 16-dword compare
])
def test_probe_kernel_choice_is_exact(pd, k, mode, monkeypatch):
    """dev_probe picks between the fast kernel with the in-kernel state machine for inconsistent reads (TAIL = 1) and the one that lists such
    reads (TAIL = 0) from a sample of the range; listed reads go through the TAIL = 1 kernel as an id list, its leftovers through the
    sequential kernel; TAIL = 2 sends every read through the state machine.  Every route gives the oracle's records: the sampled route
    (forced on a small input), and each kernel alone."""
    if mode == "sample":
        monkeypatch.setenv("SAGE2OV_PROBE_SAMPLE_MIN", "2048")
    else:
        monkeypatch.setenv("SAGE2OV_PROBE_TAIL", "2" if mode.startswith("tail2") else mode[-1])
        if mode == "tail2_own_hit_lists":                                            # TAIL = 2 writes its hits out for the reduce phase: here the reduce phase makes its own
            monkeypatch.setenv("SAGE2OV_NO_PREHITS", "1")
    bases, off = fx.make_reads(pd)
    m = dict(k=k)
    g, o = run_gpu(m, bases, off), run_oracle(m, bases, off)
    assert_equals_oracle(g, o)
    tm = g.timings()
    if mode == "sample":
        assert tm.probe_fast_launches >= 2 * tm.probe_kernel_launches            # a sample launch and the rest, per pass
    g.close(); o.close()


@pytest.mark.gpu
def test_minimiser_groups_by_the_run_start_rule_change_nothing_but_time(monkeypatch):
    """dev_build_index builds the minimiser groups when enough reads have another minimiser (or strand of it) than the read before them in the locality order -- low
    coverage: 2.3 M reads at 8x here, above the rule's 2 M-read floor -- and not otherwise.  Built by the rule, forced off and forced on: the same edge list and counters."""
    import zlib
    p = fx.synth_params(dict(seed=17, genome_len=45_000_000, n_reads=2_400_000, read_len=150))
    genome = s2.synth_genome(p); res = []
    for mode in (None, "0", "1"):
        if mode is None:
            monkeypatch.delenv("SAGE2OV_MINIMIZER_INDEX", raising=False)
        else:
            monkeypatch.setenv("SAGE2OV_MINIMIZER_INDEX", mode)
        c = s2.Context(40, device=0); c.reads_add_synth(p, genome); c.reads_organize(); c.run_steps23()
        e = c.edges(); st = c.overlap_stats()
        res.append((zlib.crc32(e.tobytes()), len(e), st.verified_overlaps, st.contained_extension, st.edges)); c.close()
    assert res[0] == res[1] == res[2] and res[0][1] > 1_000_000


@pytest.mark.gpu
@pytest.mark.parametrize("chunk_shift", ["7", "8"])
@pytest.mark.parametrize("phase_blocks", ["1", "3", "7"])
@pytest.mark.parametrize("pd,k", [
    (dict(seed=81, genome_len=60000, n_reads=20000, read_len=150, err_ppm=0), 40),
    (dict(seed=82, genome_len=45000, n_reads=15000, read_len=150, read_len_min=100, err_ppm=700), 33),
])
def test_tapered_grid_of_the_fast_kernel_is_exact(pd, k, phase_blocks, chunk_shift, monkeypatch):
    """Big launches of the fast probe kernel run a TAPERED grid (plan_fast_grid: up to three phases of 4096 blocks that take 3/4 of the remaining chunks each, then short
    blocks -- the launch has no long tail); SAGE2OV_TEST_PHASE_BLOCKS shrinks a phase to a few blocks so that a few thousand reads walk through all four phases.
    Every chunk must be visited exactly once: the oracle's records, counters and edges.  A block takes 128 positions per visit, 256 in launches over 24 M positions
    and more (ProbeArgs::chunkShift; SAGE2OV_FAST_CHUNK_SHIFT forces either size here)."""
    monkeypatch.setenv("SAGE2OV_TEST_PHASE_BLOCKS", phase_blocks)
    monkeypatch.setenv("SAGE2OV_FAST_CHUNK_SHIFT", chunk_shift)
    monkeypatch.setenv("SAGE2OV_PROBE_TAIL", "1")                                   # (one launch over the whole range with the kernel that settles inconsistent reads itself)
    bases, off = fx.make_reads(pd)
    m = dict(k=k)
    g, o = run_gpu(m, bases, off), run_oracle(m, bases, off)
    assert_equals_oracle(g, o)
    g.close(); o.close()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["wide_pass", "sequential_form", "both", "standard_form", "sampled_route"])
@pytest.mark.parametrize("pd,k", [
    (dict(seed=91, genome_len=16000, n_reads=16000, read_len=150, err_ppm=0), 40),       # 150x: most reads have 129 .. 256 candidates
    (dict(seed=92, genome_len=9000, n_reads=14000, read_len=100, err_ppm=0), 31),        # 155x of 100-base reads (4-word layout)
    (dict(seed=93, genome_len=20000, n_reads=16000, read_len=150, err_ppm=300), 40),     # with a few read errors: inconsistent reads go on to the state machine
    (dict(seed=94, genome_len=8000, n_reads=20000, read_len=150, err_ppm=0), 40),        # 375x: beyond 256 candidates, the sequential kernel as before
])
def test_sequential_groups_form_and_wide_pass_are_exact(pd, k, mode, monkeypatch):
    """The clean-data probe kernel has a form that gathers and compares its 64-slot candidate groups one after the other (k_probe_fast<..., UNI, QN, SEQ>): with two
    groups it replaces the standard form (SAGE2OV_PROBE_SEQ), with four it takes the reads of high-coverage data that the 128-slot forms list (more than 128 candidates)
    before anything goes to the state machine or the sequential kernel.  Same records, counters and edges as the oracle either way."""
    if mode in ("wide_pass", "both"):
        monkeypatch.setenv("SAGE2OV_PROBE_WIDE_MIN", "1")
    elif mode != "sampled_route":
        monkeypatch.setenv("SAGE2OV_NO_WIDE", "1")
    monkeypatch.setenv("SAGE2OV_PROBE_SEQ", "1" if mode in ("sequential_form", "both", "sampled_route") else "0")   # (0: the standard form, both groups in flight together)
    if mode == "sampled_route":                                                     # the host's own route: a sample counts the reads with too many candidates and picks the wide form for the rest
        monkeypatch.setenv("SAGE2OV_PROBE_SAMPLE_MIN", "1024"); monkeypatch.setenv("SAGE2OV_PROBE_WIDE_MIN", "1")
    else:
        monkeypatch.setenv("SAGE2OV_PROBE_TAIL", "0")                               # (the form that lists what it cannot settle: the wide pass works on that list)
    bases, off = fx.make_reads(pd)
    m = dict(k=k)
    g, o = run_gpu(m, bases, off), run_oracle(m, bases, off)
    assert_equals_oracle(g, o)
    g.close(); o.close()


@pytest.mark.gpu
def test_reads_beyond_the_longest_layout_are_refused():
    """1018 bases is the limit of the 32-word layout; a longer read is reported by the organiser (ASCII and device-pack path alike), not truncated."""
    pd = dict(seed=5, genome_len=20000, n_reads=300, read_len=1019, read_len_min=900)
    bases, off = fx.make_reads(pd)
    ctx = s2.Context(31)
    ctx.reads_add_ascii(bases, off)
    with pytest.raises(Exception) as ei:
        ctx.reads_organize()
    assert "1018" in str(ei.value)
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("pd,k", [
    (dict(seed=91, genome_len=60000, n_reads=24000, read_len=150, err_ppm=0), 40),
    (dict(seed=92, genome_len=60000, n_reads=24000, read_len=150, err_ppm=1500), 40),
    (dict(seed=93, genome_len=40000, n_reads=16000, read_len=100, read_len_min=70, err_ppm=1000, n_repeat_families=3, repeat_copies=40, repeat_len=300), 21),
    (dict(seed=94, genome_len=50000, n_reads=10000, read_len=250, err_ppm=800), 55),
])
def test_without_minimiser_groups_matches_oracle(pd, k, monkeypatch):
    """what the library does by itself on small inputs (and on every rank of a multi-GPU run): no minimiser groups, every window that window reuse
    leaves open probes the uniform table"""
    monkeypatch.setenv("SAGE2OV_MINIMIZER_INDEX", "0")
    bases, off = fx.make_reads(pd)
    m = dict(k=k)
    g, o = run_gpu(m, bases, off), run_oracle(m, bases, off)
    assert_equals_oracle(g, o)
    g.close(); o.close()


@pytest.mark.gpu
@pytest.mark.parametrize("pd,k", [
    (dict(seed=16388, genome_len=73072, n_reads=28823, read_len=161, read_len_min=150, err_ppm=1500, n_repeat_families=1, repeat_copies=233, repeat_len=136), 40),
    (dict(seed=16461, genome_len=29064, n_reads=26966, read_len=161, err_ppm=1500, n_repeat_families=2, repeat_copies=161, repeat_len=348), 70),
    (dict(seed=301, genome_len=60000, n_reads=24000, read_len=200, err_ppm=1500, n_repeat_families=3, repeat_copies=150, repeat_len=300), 31),
    (dict(seed=302, genome_len=50000, n_reads=20000, read_len=250, read_len_min=170, err_ppm=3000, n_repeat_families=2, repeat_copies=250, repeat_len=200), 55),
])
def test_mid_length_reads_in_repeats_match_oracle(pd, k, monkeypatch):
    """161-251-bp reads (16-dword instantiations of the fast kernel) with read errors and high-copy repeats, device reduce forced: the hit-list
    kernel of these layouts returned wrong overhang lengths when it was compiled with spilled registers (the first two data
    sets are the ones the stress generator found it with)."""
    monkeypatch.setenv("SAGE2OV_DEVICE_REDUCE_MIN", "1")
    bases, off = fx.make_reads(pd)
    m = dict(k=k)
    g, o = run_gpu(m, bases, off), run_oracle(m, bases, off)
    assert_equals_oracle(g, o)
    g.close(); o.close()


@pytest.mark.parametrize("name", ["g2_clean150_k40", "g3_noisy_rep_k21", "g4_highcopy_k21", "g5_mixedlen_k21"])
def test_memory_diet_mode_matches_reference(name, tmp_path, monkeypatch):
    """The mode a billion-read context runs in (BASELINE configs[4]): transient buffers released at the end of their phase, the id-ordered
    read store released once the locality-ordered one exists (uniform lengths only: g5 keeps it), no minimiser groups, per-read results and the candidate
    list allocated when first needed and sized by a counting pass.  Same files as the reference's, four times in a row (the second index build starts without
    the id-ordered store; the phases' buffers are carved out of one block that the first two steps size and the later ones only reuse: sage2ov_device.hip,
    Device::Phase), and the arena is smaller than the default mode's."""
    m = fx.golden(name)
    bases, off = fx.make_reads(m["synth"])
    def run(diet):
        monkeypatch.setenv("SAGE2OV_MEMORY_DIET", "1" if diet else "0")
        c = s2.Context(m["k"], device=0)
        c.reads_add_ascii(bases, off); c.reads_organize()
        for rep in range(4 if diet else 2):
            c.run_steps23()
            gp = str(tmp_path / f"d{int(diet)}{rep}.graph3"); c.graph_save(gp)
            assert fx.graph3_matches(gp, name), f"diet={diet} pass {rep}"
        rp = str(tmp_path / f"d{int(diet)}.reads"); c.reads_save(rp); assert fx.md5_file(rp) == m["reads_md5"]
        arena = c.debug_meminfo()["arena"]; c.close()
        return arena
    assert run(True) < run(False)


@pytest.mark.gpu
@pytest.mark.parametrize("run_mode", ["on", "off"])
@pytest.mark.parametrize("pd,k", [
    (dict(seed=171, genome_len=90000, n_reads=30000, read_len=150, err_ppm=0), 40),                            # 50x, clean: nearly every read of a run is answered off the frame
    (dict(seed=172, genome_len=90000, n_reads=30000, read_len=150, err_ppm=60), 40),                           # a read error now and then: runs end on a candidate that does not verify and start again
    (dict(seed=173, genome_len=40000, n_reads=40000, read_len=100, err_ppm=0), 21),                            # 100x of 100-base reads, 80 windows, the 4-word layout; many reads past 128 slots (wide form: 256-slot ring)
    (dict(seed=174, genome_len=200000, n_reads=16000, read_len=150, err_ppm=0), 55),                           # 12x: short runs, long shifts (beyond RUN_DMAX), k = 55
    (dict(seed=175, genome_len=60000, n_reads=20000, read_len=150, err_ppm=0, n_repeat_families=3, repeat_len=300, repeat_copies=5), 40),   # repeats: candidates that share a key but not the frame
    (dict(seed=176, genome_len=60000, n_reads=20000, read_len=150, err_ppm=0), 64),                            # k = 64: the widest key whose gates do not depend on the window
    (dict(seed=177, genome_len=60000, n_reads=20000, read_len=150, err_ppm=0), 70),                            # k = 70 > 64: no run mode (hash string of 64, gates by window): the general path only
    (dict(seed=178, genome_len=9000, n_reads=30000, read_len=150, err_ppm=0), 40),                             # 500x: duplicates collapse, every window a bucket of many entries, rings fill up
])
def test_run_mode_of_the_fast_kernel_is_exact(pd, k, run_mode, monkeypatch):
    """Run mode (kernels_probe_fast.inc: the reads of a run of the locality order answered off a verified frame and a ring of verified slots) on and off
    (SAGE2OV_NO_RUN_MODE): the oracle's extension records, connection counts, edges and counters either way; and with the probe sampled on these small inputs,
    so that the sample launch, the rest and the listed reads all run their forms."""
    monkeypatch.delenv("SAGE2OV_MINIMIZER_INDEX", raising=False)          # (the library's own route on small inputs: the uniform table; the groups' route is the module default elsewhere)
    monkeypatch.setenv("SAGE2OV_PROBE_SAMPLE_MIN", "2048")
    if run_mode == "off":
        monkeypatch.setenv("SAGE2OV_NO_RUN_MODE", "1")
    bases, off = fx.make_reads(pd)
    m = dict(k=k)
    g, o = run_gpu(m, bases, off), run_oracle(m, bases, off)
    assert_equals_oracle(g, o)
    g.run_steps23(); assert_equals_oracle(g, o)                            # (a second pass over the resident reads)
    g.close(); o.close()


@pytest.mark.gpu
@pytest.mark.parametrize("form", ["parallel", "steps"])
@pytest.mark.parametrize("tail", ["1", "2"])
@pytest.mark.parametrize("pd,k", [
    (dict(seed=181, genome_len=90000, n_reads=30000, read_len=150, err_ppm=1000), 40),                          # 50x, 0.1 % errors: a rejected hit or two per side, two to three rounds
    (dict(seed=182, genome_len=90000, n_reads=30000, read_len=150, err_ppm=8000), 40),                          # 0.8 % errors: most hits rejected, rounds beyond the limit -> the steps
    (dict(seed=183, genome_len=60000, n_reads=20000, read_len=150, read_len_min=110, err_ppm=1500), 31),        # mixed lengths: `longer` decides inside a window
    (dict(seed=184, genome_len=30000, n_reads=30000, read_len=100, err_ppm=1000), 21),                          # 100x of 100-base reads: sides beyond 32 hits take the steps, the 4-word layout
    (dict(seed=185, genome_len=60000, n_reads=20000, read_len=150, err_ppm=500, n_repeat_families=3, repeat_len=300, repeat_copies=8), 40),   # repeats: hits that disagree over their whole length
    (dict(seed=186, genome_len=120000, n_reads=16000, read_len=123, err_ppm=2000), 25),                         # 16x, the 8-dword layout's longest read
    (dict(seed=187, genome_len=80000, n_reads=16000, read_len=250, err_ppm=1000), 55),                          # the 16-dword layout: overhangs of up to 16 dwords per lane
    (dict(seed=188, genome_len=80000, n_reads=20000, read_len=200, read_len_min=165, err_ppm=1500), 31),        # ... with mixed lengths
])
def test_state_machine_in_its_parallel_form_is_exact(pd, k, tail, form, monkeypatch):
    """The state machine of the fast kernel (economyGraph.cpp:95-438) evaluated as a fixed point over all hits of a side at once (kernels_probe_fast.inc: THE PARALLEL
    FORM) and step by step (SAGE2OV_NO_PAR_TAIL), in the instantiation that sends every read through it (2) and in the one that sends the inconsistent ones (1): the
    oracle's extension records, connection counts, edges and counters every way."""
    monkeypatch.setenv("SAGE2OV_PROBE_TAIL", tail)
    if form == "steps":
        monkeypatch.setenv("SAGE2OV_NO_PAR_TAIL", "1")
    bases, off = fx.make_reads(pd)
    m = dict(k=k)
    g, o = run_gpu(m, bases, off), run_oracle(m, bases, off)
    assert_equals_oracle(g, o)
    g.close(); o.close()


@pytest.mark.gpu
def test_a_key_in_more_than_2_to_20_reads(monkeypatch):
    """One k-mer as the prefix of 1.15 M different reads (an adaptor, a low-complexity stretch at human scale): the bucket's counter in the window kernel has 20 bits by
    default; the build notices the overflow and runs again with 30 bits of count instead of failing with SAGE2OV_ERR_LIMIT.  The bucket is long
    (hashTable.cpp:111-123: never found), its reads find each other through their other keys or not at all: the oracle's records, edges and counters."""
    rng = np.random.default_rng(20260401)
    n, L, k = 1_150_000, 64, 21
    pre = np.frombuffer(b"ACGTTGCAAGGCTTACGATCC", dtype=np.uint8)
    tail = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(n, L - len(pre)))]
    bases = np.concatenate([np.broadcast_to(pre, (n, len(pre))), tail], axis=1).reshape(-1).copy()
    off = (np.arange(n + 1, dtype=np.uint64) * np.uint64(L))
    m = dict(k=k)
    g, o = run_gpu(m, bases, off), run_oracle(m, bases, off)
    assert g.index_stats().long_buckets >= 1
    assert_equals_oracle(g, o)
    g.close(); o.close()
