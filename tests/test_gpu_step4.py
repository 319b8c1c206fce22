"""GPU: step 4 (graph simplification on the device, through the C ABI) against the graphs the reference's own classes hold after their
step 4 (tests/golden/*.graph4.gz, dumped by oracle/ref_driver.cpp::sage2ref_run_step4) and against the CPU restatement
(oracle/step4_oracle.cpp) on larger inputs: byte identity of the written file -- edges, read lists and their order."""
import ctypes, gzip, json, os, shutil
import numpy as np
import pytest
import fixtures as fx
import sage2_amd as s2

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("minimiser_groups_on")]      # (small inputs: the groups' half of the look-up code is exercised by request, conftest.py)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _oracle4():
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "liboracle_step4.so"))
    lib.orc4_run_files.argtypes = [ctypes.c_char_p, ctypes.c_ulonglong, ctypes.c_char_p, ctypes.POINTER(ctypes.c_ulonglong)]
    return lib


def _first_diff(a: bytes, b: bytes):
    n = min(len(a), len(b)); x = next((i for i in range(n) if a[i] != b[i]), n)
    return f"first difference at byte {x} of {len(a)}/{len(b)}: got {a[max(0, x - 120):x + 120]!r} want {b[max(0, x - 120):x + 120]!r}"


@pytest.mark.parametrize("name", fx.golden_names())
def test_graph4_equals_reference_dump(name, tmp_path):
    m = fx.golden(name); m4 = json.load(open(os.path.join(fx.GOLDEN, name + ".step4.json")))
    bases, off = fx.make_reads(m["synth"])
    ctx = s2.Context(m["k"]); ctx.reads_add_ascii(bases, off); ctx.reads_organize(); ctx.run_steps23()
    ctx.graph_simplify()
    st = ctx.simplify_stats()
    out = str(tmp_path / "t.graph4"); ctx.graph4_save(out); ctx.close()
    want = gzip.open(os.path.join(fx.GOLDEN, name + ".graph4.gz")).read(); got = open(out, "rb").read()
    assert (st.nodes_contracted, st.removed, st.loop_iterations) == (m4["counters"]["nodes_contracted"], m4["counters"]["removed"], m4["counters"]["loop_iterations"])
    assert got == want, _first_diff(got, want)


@pytest.mark.parametrize("pd,k", [
    (dict(seed=41, genome_len=300000, n_reads=100000, read_len=150), 40),
    (dict(seed=42, genome_len=300000, n_reads=100000, read_len=150, err_ppm=2000), 40),
    (dict(seed=43, genome_len=200000, n_reads=80000, read_len=100, err_ppm=3000, n_repeat_families=6, repeat_copies=8, repeat_len=400), 21),
    (dict(seed=44, genome_len=150000, n_reads=80000, read_len=120, read_len_min=70, err_ppm=5000), 25),
])
def test_graph4_equals_restatement(pd, k, tmp_path):
    bases, off = fx.make_reads(pd)
    ctx = s2.Context(k); ctx.reads_add_ascii(bases, off); ctx.reads_organize(); ctx.run_steps23()
    g3 = str(tmp_path / "t.graph3"); ctx.graph_save(g3)
    ctx.graph_simplify(); st = ctx.simplify_stats()
    out = str(tmp_path / "t.graph4"); ctx.graph4_save(out)
    n = ctx.reads_stats().unique_reads; ctx.close()
    c = (ctypes.c_ulonglong * 5)(); ref = str(tmp_path / "ref.graph4")
    assert _oracle4().orc4_run_files(g3.encode(), n, ref.encode(), c) == 0
    assert (st.nodes_contracted, st.removed, st.loop_iterations) == (c[2], c[3], c[1])
    got, want = open(out, "rb").read(), open(ref, "rb").read()
    assert got == want, _first_diff(got, want)


@pytest.mark.parametrize("seed", range(40))
def test_graph4_on_synthetic_graphs(seed, tmp_path):
    """graphs the read pipeline never produces on random genomes: cycles, closed chains, parallel chains with and without a direct edge,
    two-cycles, nodes whose edges do not combine, multi-edges (tests/graphgen.py), imported through sage2ov_edges_import"""
    import graphgen as gg
    N, e = gg.random_graph(seed, n_anchor=8 + 3 * seed, n_paths=20 + 6 * seed, max_len=3 + seed % 9, n_cycles=seed % 4, p_bad=0.02 * (seed % 3))
    rng = np.random.default_rng(seed)
    reads = set()
    while len(reads) < N:
        reads.add("".join(rng.choice(list("ACGT"), size=100)))
    bases = np.frombuffer("".join(sorted(reads)).encode(), dtype=np.uint8).copy(); off = np.arange(0, (N + 1) * 100, 100, dtype=np.uint64)
    ctx = s2.Context(40); ctx.reads_add_ascii(bases, off); ctx.reads_organize()
    assert ctx.reads_stats().unique_reads == N
    ctx.edges_import(e)
    g3 = str(tmp_path / "t.graph3"); ctx.graph_save(g3)
    ctx.graph_simplify(); st = ctx.simplify_stats()
    out = str(tmp_path / "t.graph4"); ctx.graph4_save(out); ctx.close()
    c = (ctypes.c_ulonglong * 5)(); ref = str(tmp_path / "ref.graph4")
    assert _oracle4().orc4_run_files(g3.encode(), N, ref.encode(), c) == 0
    got, want = open(out, "rb").read(), open(ref, "rb").read()
    assert (st.nodes_contracted, st.removed, st.loop_iterations) == (c[2], c[3], c[1]), _first_diff(got, want)
    assert got == want, _first_diff(got, want)


def test_graph4_one_million_noisy_reads(tmp_path):
    """about a million nodes, chains of a few nodes between error branches, several rounds of the loop: against the restatement"""
    pd = dict(seed=45, genome_len=3000000, n_reads=1000000, read_len=150, err_ppm=1000)
    bases, off = fx.make_reads(pd)
    ctx = s2.Context(40); ctx.reads_add_ascii(bases, off); ctx.reads_organize(); ctx.run_steps23()
    g3 = str(tmp_path / "t.graph3"); ctx.graph_save(g3)
    ctx.graph_simplify(); st = ctx.simplify_stats()
    out = str(tmp_path / "t.graph4"); ctx.graph4_save(out)
    n = ctx.reads_stats().unique_reads; ctx.close()
    c = (ctypes.c_ulonglong * 5)(); ref = str(tmp_path / "ref.graph4")
    assert _oracle4().orc4_run_files(g3.encode(), n, ref.encode(), c) == 0
    assert (st.nodes_contracted, st.removed, st.loop_iterations) == (c[2], c[3], c[1])
    assert fx.md5_file(out) == fx.md5_file(ref)


def test_cli_steps_1_to_4_and_restart_at_4(tmp_path):
    """`sage2ov -M 4` writes P.graph4 (and P.graph3 with -s); `sage2ov -m 4` on the saved P.reads + P.graph3 writes the same P.graph4"""
    import subprocess
    name = "g3_noisy_rep_k21"; m = fx.golden(name)
    fa = str(tmp_path / "x.fa"); s2.synth_write_fasta(fx.synth_params(m["synth"]), fa)
    exe = os.path.join(ROOT, "sage2_amd", "sage2ov"); out = str(tmp_path / "out")
    subprocess.run([exe, "-f", fa, "-k", str(m["k"]), "-o", out, "-p", "t", "-M", "4", "-s"], check=True, stdout=subprocess.DEVNULL)
    want = gzip.open(os.path.join(fx.GOLDEN, name + ".graph4.gz")).read()
    assert open(os.path.join(out, "t.graph3"), "rb").read() == fx.golden_graph3(name)
    assert open(os.path.join(out, "t.graph4"), "rb").read() == want
    subprocess.run([exe, "-k", str(m["k"]), "-o", out, "-p", "u", "-i", "t", "-m", "4", "-M", "4"], check=True, stdout=subprocess.DEVNULL)
    assert open(os.path.join(out, "u.graph4"), "rb").read() == want
    # the restart reads P.reads and P.graph3 with the mapped, chunk-parallel loaders; the one-thread loaders (SAGE2OV_SEQUENTIAL_READER) give the same file
    subprocess.run([exe, "-k", str(m["k"]), "-o", out, "-p", "w", "-i", "t", "-m", "4", "-M", "4"], check=True, stdout=subprocess.DEVNULL, env=dict(os.environ, SAGE2OV_SEQUENTIAL_READER="1"))
    assert open(os.path.join(out, "w.graph4"), "rb").read() == want
    # a P.graph3 that is not strictly of the writers' shape (blanks instead of tabs) still loads: the general parser takes it
    txt = open(os.path.join(out, "t.graph3")).read().replace("\t", " ")
    open(os.path.join(out, "b.graph3"), "w").write(txt); shutil.copy(os.path.join(out, "t.reads"), os.path.join(out, "b.reads"))
    subprocess.run([exe, "-k", str(m["k"]), "-o", out, "-p", "x", "-i", "b", "-m", "4", "-M", "4"], check=True, stdout=subprocess.DEVNULL)
    assert open(os.path.join(out, "x.graph4"), "rb").read() == want


def test_plain_jumping_path_equals_splitter_ranking(monkeypatch, tmp_path):
    """the chain ranking has two forms (splitter-based O(n); plain pointer jumping as the fallback for cycles without a splitter and
    for over-long stretches): same file from both"""
    import graphgen as gg
    outs = {}
    for mode in ("default", "plain"):
        if mode == "plain":
            monkeypatch.setenv("SAGE2OV_S4_PLAIN_JUMPING", "1")
        N, e = gg.random_graph(77, n_anchor=200, n_paths=700, max_len=90, n_cycles=12, p_bad=0.01)
        rng = np.random.default_rng(77); reads = set()
        while len(reads) < N:
            reads.add("".join(rng.choice(list("ACGT"), size=100)))
        bases = np.frombuffer("".join(sorted(reads)).encode(), dtype=np.uint8).copy(); off = np.arange(0, (N + 1) * 100, 100, dtype=np.uint64)
        ctx = s2.Context(40); ctx.reads_add_ascii(bases, off); ctx.reads_organize(); ctx.edges_import(e)
        g3 = str(tmp_path / f"{mode}.graph3"); ctx.graph_save(g3)
        ctx.graph_simplify(); out = str(tmp_path / f"{mode}.graph4"); ctx.graph4_save(out); ctx.close()
        outs[mode] = open(out, "rb").read()
    c = (ctypes.c_ulonglong * 5)(); ref = str(tmp_path / "ref.graph4")
    assert _oracle4().orc4_run_files(g3.encode(), N, ref.encode(), c) == 0
    want = open(ref, "rb").read()
    assert outs["default"] == want, _first_diff(outs["default"], want)
    assert outs["plain"] == want, _first_diff(outs["plain"], want)


def _run_imported(N, e, tmp_path, tag):
    rng = np.random.default_rng(N); reads = set()
    while len(reads) < N:
        reads.add("".join(rng.choice(list("ACGT"), size=60)))
    bases = np.frombuffer("".join(sorted(reads)).encode(), dtype=np.uint8).copy(); off = np.arange(0, (N + 1) * 60, 60, dtype=np.uint64)
    ctx = s2.Context(40); ctx.reads_add_ascii(bases, off); ctx.reads_organize(); ctx.edges_import(e)
    g3 = str(tmp_path / f"{tag}.graph3"); ctx.graph_save(g3)
    ctx.graph_simplify(); st = ctx.simplify_stats(); out = str(tmp_path / f"{tag}.graph4"); ctx.graph4_save(out); ctx.close()
    c = (ctypes.c_ulonglong * 5)(); ref = str(tmp_path / f"{tag}.ref4")
    assert _oracle4().orc4_run_files(g3.encode(), N, ref.encode(), c) == 0
    got, want = open(out, "rb").read(), open(ref, "rb").read()
    assert (st.nodes_contracted, st.removed, st.loop_iterations) == (c[2], c[3], c[1])
    assert got == want, _first_diff(got, want)
    return st


def _ring(n, seed, closed):
    """n nodes in one path (or one cycle) of forward-forward overlaps, ids scattered"""
    import graphgen as gg
    rng = np.random.default_rng(seed); perm = rng.permutation(n) + 1
    rows = []
    for j in range(n if closed else n - 1):
        a, b = int(perm[j]), int(perm[(j + 1) % n]); t = 3
        if a > b: a, b, t = b, a, 0
        rows.append((a, b, t, int(rng.integers(1, 100))))
    rows.sort()
    e = np.zeros(len(rows), dtype=gg.EDGE_DTYPE)
    for i, (a, b, t, ln) in enumerate(rows):
        e[i]["from"], e[i]["to"], e[i]["type"], e[i]["length"], e[i]["length_twin"] = a, b, t, ln, ln
    return e


def test_one_long_chain_and_one_big_cycle(tmp_path):
    """a linear genome is one chain (everything but the two ends is contracted), a circular one is a cycle without any branching node:
    its three largest ids survive (simplification.cpp:27-34 stops the sweep at a triangle)"""
    n = 200000
    st = _run_imported(n, _ring(n, 5, closed=False), tmp_path, "chain")
    assert st.nodes_contracted == n - 2 and st.edges == 1
    st = _run_imported(n, _ring(n, 6, closed=True), tmp_path, "cycle")
    assert st.nodes_contracted == n - 3 and st.edges == 3


def test_graph_without_edges(tmp_path):
    import graphgen as gg
    st = _run_imported(500, np.zeros(0, dtype=gg.EDGE_DTYPE), tmp_path, "empty")
    assert st.edges == 0 and st.nodes_contracted == 0


def test_hub_with_thousands_of_edges(tmp_path):
    """a node with 3000 edges (a collapsed repeat): its list is heap-sorted; tips of one and two nodes around it"""
    import graphgen as gg
    rng = np.random.default_rng(9); n_spokes = 3000
    N = 1 + 2 * n_spokes; perm = rng.permutation(N) + 1; hub = int(perm[0]); rows = set()
    for j in range(n_spokes):
        a, b = int(perm[1 + 2 * j]), int(perm[2 + 2 * j])
        t1 = int(rng.integers(0, 4)); x, y, t = (hub, a, t1) if hub < a else (a, hub, {0: 3, 3: 0, 1: 1, 2: 2}[t1])
        rows.add((x, y, t, int(rng.integers(1, 60))))
        if j % 3:
            t2 = int(rng.integers(0, 4)); x, y, t = (a, b, t2) if a < b else (b, a, {0: 3, 3: 0, 1: 1, 2: 2}[t2])
            rows.add((x, y, t, int(rng.integers(1, 60))))
    rows = sorted(rows)
    e = np.zeros(len(rows), dtype=gg.EDGE_DTYPE)
    for i, (a, b, t, ln) in enumerate(rows):
        e[i]["from"], e[i]["to"], e[i]["type"], e[i]["length"], e[i]["length_twin"] = a, b, t, ln, ln
    _run_imported(N, e, tmp_path, "hub")


def test_full_size_properties_c2(tmp_path):
    """BASELINE configs[1] at full size, step 4: the reduced graph of an error-free chromosome is one path, so the sweep contracts every
    read but the two ends into ONE edge pair; size-independent properties of the written file: both read lists hold every contracted
    read exactly once, the twin's list is the forward list reversed with the orientations flipped, distances mirror, the edge lengths
    are the sums of the steps; a second run gives the same bytes."""
    import zlib
    pd = dict(seed=2, genome_len=30_000_000, n_reads=10_000_000, read_len=150)
    p = fx.synth_params(pd)
    ctx = s2.Context(40, device=0)
    ctx.reads_add_synth(p, s2.synth_genome(p)); ctx.reads_organize(); ctx.run_steps23()
    n = ctx.reads_stats().unique_reads
    ctx.graph_simplify(); st = ctx.simplify_stats()
    assert (st.nodes_contracted, st.removed, st.edges, st.reads_on_edges) == (n - 2, 0, 1, n - 2)
    out = str(tmp_path / "t.graph4"); ctx.graph4_save(out)
    raw = open(out, "rb").read(); crc = zlib.crc32(raw)
    lines = raw.split(b"\n")
    hdr = lines[3].split(b"\t"); a, b, cnt = int(hdr[0]), int(hdr[1]), int(hdr[6])
    assert cnt == n - 2 and a < b
    fwd = np.loadtxt(lines[4:4 + cnt], dtype=np.int64).reshape(cnt, 5)
    k2 = 4 + cnt + 1                                             # blank line, then the twin
    hdr2 = lines[k2].split(b"\t"); assert (int(hdr2[0]), int(hdr2[1]), int(hdr2[6])) == (b, a, cnt)
    rev = np.loadtxt(lines[k2 + 1:k2 + 1 + cnt], dtype=np.int64).reshape(cnt, 5)
    ids = np.sort(fwd[:, 0]); assert np.array_equal(ids, np.setdiff1d(np.arange(1, n + 1), [a, b]))
    assert np.array_equal(rev[::-1, 0], fwd[:, 0]) and np.array_equal(rev[::-1, 1], 1 - fwd[:, 1])
    assert np.array_equal(rev[::-1, 3], fwd[:, 4]) and np.array_equal(rev[::-1, 4], fwd[:, 3])          # equal read lengths: a step has one length both ways
    assert np.array_equal(fwd[1:, 3], fwd[:-1, 4])                                                      # distPrevious of a read = distNext of the one before
    assert int(hdr[4]) == int(fwd[:, 3].sum() + fwd[-1, 4]) and int(hdr2[4]) == int(hdr[4])
    ctx.graph_simplify(); out2 = str(tmp_path / "u.graph4"); ctx.graph4_save(out2)
    assert zlib.crc32(open(out2, "rb").read()) == crc
    ctx.close()


def test_cli_end_to_end_fastq_250bp_against_both_oracles(tmp_path):
    """one run through everything: a 5 MB four-line FASTQ of 250-bp reads with errors (chunk-parallel reader, 8-word slots with the 16-dword
    compare, sequential-kernel hand-overs, device reduce), `sage2ov -M 4 -s`; P.reads / P.graph3 against the steps 1-3 oracle, P.graph4
    against the step-4 oracle"""
    import subprocess
    import oracle_lib as ol
    pd = dict(seed=61, genome_len=100000, n_reads=20000, read_len=250, err_ppm=1500)
    bases, off = fx.make_reads(pd)
    fq = str(tmp_path / "x.fq")
    with open(fq, "w") as f:
        for i in range(len(off) - 1):
            s = bytes(bases[int(off[i]):int(off[i + 1])]).decode()
            f.write(f"@r{i}\n{s}\n+\n{'@' * len(s)}\n")
    assert os.path.getsize(fq) > (1 << 20)
    exe = os.path.join(ROOT, "sage2_amd", "sage2ov"); out = str(tmp_path / "out")
    subprocess.run([exe, "-f", fq, "-k", "45", "-o", out, "-p", "t", "-M", "4", "-s"], check=True, stdout=subprocess.DEVNULL)
    o = ol.Oracle(45, 8); o.add_reads_ascii(bases, off); o.organize(); o.run_all()
    o.write_reads(str(tmp_path / "o.reads")); o.write_graph3(str(tmp_path / "o.graph3")); n = o.counter("N"); o.close()
    assert open(os.path.join(out, "t.reads"), "rb").read() == open(tmp_path / "o.reads", "rb").read()
    assert open(os.path.join(out, "t.graph3"), "rb").read() == open(tmp_path / "o.graph3", "rb").read()
    c = (ctypes.c_ulonglong * 5)(); ref = str(tmp_path / "ref.graph4")
    assert _oracle4().orc4_run_files(str(tmp_path / "o.graph3").encode(), n, ref.encode(), c) == 0
    got, want = open(os.path.join(out, "t.graph4"), "rb").read(), open(ref, "rb").read()
    assert got == want, _first_diff(got, want)
