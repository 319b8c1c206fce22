"""CPU: the C-ABI library loads and exports every symbol include/sage2ov.h declares; the step-1 host
path (which needs no GPU) reproduces the reference's P.reads; compute calls fail loudly without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import fixtures as fx
import sage2_amd as s2

HEADER = os.path.join(fx.ROOT, "include", "sage2ov.h")


def declared_symbols():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(sage2ov_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    L = s2.lib()
    syms = declared_symbols()
    assert len(syms) >= 35
    for s in syms:
        assert hasattr(L, s), f"{s} declared in sage2ov.h but not exported"
    assert b"gfx950" in L.sage2ov_version()


def test_step1_host_path_reproduces_reference_reads_file(tmp_path):
    for name in ("g1_clean100_k21", "g5_mixedlen_k21", "g6_k70_150"):
        m = fx.golden(name)
        bases, off = fx.make_reads(m["synth"])
        ctx = s2.Context(m["k"], device=-2)            # SAGE2OV_DEVICE_NONE: step 1 only
        ctx.reads_add_ascii(bases, off)
        ctx.reads_organize()
        st = ctx.reads_stats()
        assert st.unique_reads == m["counters"]["unique_reads"] and st.good_reads == m["counters"]["good_reads"]
        p = str(tmp_path / (name + ".reads"))
        ctx.reads_save(p)
        assert fx.md5_file(p) == m["reads_md5"]
        ctx.close()


def test_fasta_reader_and_synth_paths_agree(tmp_path):
    m = fx.golden("g2_clean150_k40")
    p = fx.synth_params(m["synth"])
    fa = str(tmp_path / "x.fa")
    s2.synth_write_fasta(p, fa)
    assert fx.md5_file(fa) == m["fasta_md5"]
    a = s2.Context(m["k"], device=-2); a.reads_add_file(fa); a.reads_organize()
    b = s2.Context(m["k"], device=-2); b.reads_add_synth(p, s2.synth_genome(p)); b.reads_organize()
    pa, la, fa_ = a.reads_export(); pb, lb, fb = b.reads_export()
    assert np.array_equal(pa, pb) and np.array_equal(la, lb) and np.array_equal(fa_, fb)
    # a list file (readLoader.cpp:86-118) with split mates gives the same set
    f1, f2 = str(tmp_path / "m1.fa"), str(tmp_path / "m2.fa")
    with open(fa) as f, open(f1, "w") as o1, open(f2, "w") as o2:
        lines = f.read().splitlines()
        for r in range(0, len(lines), 2):
            (o1 if (r // 2) % 2 == 0 else o2).write(lines[r] + "\n" + lines[r + 1] + "\n")
    lst = str(tmp_path / "in.list")
    open(lst, "w").write(f"# comment\nf1={f1}\nf2={f2}\n")
    c = s2.Context(m["k"], device=-2); c.reads_add_list(lst); c.reads_organize()
    pc, lc, fc = c.reads_export()
    assert np.array_equal(pa, pc) and np.array_equal(fa_, fc)
    for x in (a, b, c):
        x.close()


def test_filters_and_errors():
    ctx = s2.Context(5, device=-2)
    reads = [b"ACGTAC", b"ACGTN", b"ACGT", b"acgtacgt", b"GTACGT"]   # N -> bad; len<=k -> small; lower case folded
    bases = np.frombuffer(b"".join(reads), dtype=np.uint8)
    off = np.cumsum([0] + [len(r) for r in reads]).astype(np.uint64)
    ctx.reads_add_ascii(bases, off)
    ctx.reads_organize()
    st = ctx.reads_stats()
    assert (st.total_reads, st.good_reads) == (5, 3)
    assert st.unique_reads == 2            # ACGTAC and GTACGT are reverse complements of each other
    with pytest.raises(s2.Sage2ovError) as e:
        ctx.index_build()
    assert e.value.code == -3              # SAGE2OV_ERR_DEVICE: no silent CPU fallback
    ctx.close()
    with pytest.raises(s2.Sage2ovError):
        s2.Context(0, device=-2)           # -k is required (main.cpp:506-510)


def test_cli_step1_and_reference_tail(tmp_path):
    """`sage2ov -M 1` (no GPU needed) writes the reference's P.reads; with the P.graph3 our GPU path is proven to
    reproduce byte for byte (tests/golden), the UNCHANGED reference continues from step 4 to contigs/scaffolds and
    gives the same result as a reference-only run.  The tail part needs oracle/_ref."""
    import shutil
    import subprocess
    cli = os.path.join(fx.ROOT, "sage2_amd", "sage2ov")
    if not os.path.exists(cli):
        pytest.skip("CLI not built")
    m = fx.golden("g2_clean150_k40")
    fa = str(tmp_path / "x.fa")
    s2.synth_write_fasta(fx.synth_params(m["synth"]), fa)
    out = str(tmp_path / "ours")
    subprocess.run([cli, "-f", fa, "-k", str(m["k"]), "-o", out, "-p", "t", "-M", "1"], check=True)
    assert fx.md5_file(os.path.join(out, "t.reads")) == m["reads_md5"]
    # usage errors print to stdout and exit 0, like the reference (main.cpp:17-24)
    r = subprocess.run([cli, "-f", fa], stdout=subprocess.PIPE)
    assert r.returncode == 0 and b"minOverlap is required" in r.stdout
    ref = os.path.join(fx.ROOT, "oracle", "_ref", "SAGE2")
    if not os.path.exists(ref):
        pytest.skip("reference binary not built (only available in the build container)")
    open(os.path.join(out, "t.graph3"), "wb").write(fx.golden_graph3("g2_clean150_k40"))
    env = dict(os.environ, OMP_NUM_THREADS="4", LC_ALL="C")
    subprocess.run([ref, "-f", fa, "-k", str(m["k"]), "-o", out, "-p", "t2", "-i", "t", "-m", "4", "-M", "7"], check=True, env=env, stdout=subprocess.DEVNULL, cwd=str(tmp_path))
    full = str(tmp_path / "full")
    subprocess.run([ref, "-f", fa, "-k", str(m["k"]), "-o", full, "-p", "t2", "-M", "7"], check=True, env=env, stdout=subprocess.DEVNULL, cwd=str(tmp_path))
    for suffix in ("_contig.fasta", "_scaffold.fasta"):
        a, b = os.path.join(out, "t2" + suffix), os.path.join(full, "t2" + suffix)
        assert os.path.exists(a) and os.path.exists(b) and open(a, "rb").read() == open(b, "rb").read()


@pytest.mark.parametrize("name", ["g1_clean100_k21", "g2_clean150_k40", "g6_k70_150"])
def test_reference_continues_from_graph4(name, tmp_path):
    """P.graph4 -- the post-step-4 graph our device step 4 is proven to reproduce byte for byte (tests/test_gpu_step4.py) -- is what the
    UNCHANGED reference's step 5 loads (main.cpp:196): `SAGE2 -m 5 -M 7 -s` on P.reads + P.graph4 writes the same P.graph5 and P.graph6
    (copy counts, mate-pair merges) as a reference-only run of all seven steps.  The final contigs are compared on the error-free
    fixtures only: on noisy ones the reference does not reproduce its own contigs across ANY restart (`-m 4` from its own P.graph3
    differs from its one-shot run in the same way, single-threaded too), although the graph files agree.  Not covered: g3/g5 (the
    reference's own steps 5-7 crash on them) and g4 (high-copy repeats: the min-cost flow has ties, which CS2 breaks by arc order, and the
    reference's loader does not restore the in-memory list order of a simplified graph -- no restart of the reference reproduces those
    flows; the graph structure of P.graph5 is the same).  Needs oracle/_ref (build container)."""
    import gzip
    import shutil
    import subprocess
    ref = os.path.join(fx.ROOT, "oracle", "_ref", "SAGE2")
    if not os.path.exists(ref):
        pytest.skip("reference binary not built (only available in the build container)")
    m = fx.golden(name)
    fa = str(tmp_path / "x.fa"); s2.synth_write_fasta(fx.synth_params(m["synth"]), fa)
    env = dict(os.environ, OMP_NUM_THREADS="4", LC_ALL="C")
    full = str(tmp_path / "full")
    subprocess.run([ref, "-f", fa, "-k", str(m["k"]), "-o", full, "-p", "t", "-M", "7", "-s"], check=True, env=env, stdout=subprocess.DEVNULL, cwd=str(tmp_path))
    out = str(tmp_path / "ours"); os.makedirs(out)
    shutil.copy(os.path.join(full, "t.reads"), os.path.join(out, "t.reads"))
    open(os.path.join(out, "t.graph4"), "wb").write(gzip.open(os.path.join(fx.GOLDEN, name + ".graph4.gz")).read())
    subprocess.run([ref, "-f", fa, "-k", str(m["k"]), "-o", out, "-p", "t2", "-i", "t", "-m", "5", "-M", "7", "-s"], check=True, env=env, stdout=subprocess.DEVNULL, cwd=str(tmp_path))
    suffixes = [".graph5", ".graph6"] + (["_contig.fasta", "_scaffold.fasta"] if "clean" in name else [])
    for suffix in suffixes:
        a, b = os.path.join(out, "t2" + suffix), os.path.join(full, "t" + suffix)
        assert os.path.exists(a) and os.path.exists(b) and open(a, "rb").read() == open(b, "rb").read(), suffix


def _export(path, k, sequential, monkeypatch):
    if sequential:
        monkeypatch.setenv("SAGE2OV_SEQUENTIAL_READER", "1")
    else:
        monkeypatch.delenv("SAGE2OV_SEQUENTIAL_READER", raising=False)
    c = s2.Context(k, device=-2); c.reads_add_file(path); c.reads_organize()
    st = c.reads_stats(); out = c.reads_export(); c.close()
    return out, (st.total_reads, st.good_reads, st.unique_reads, st.total_bp)


@pytest.mark.parametrize("form", ["fasta", "fasta_crlf", "fasta_inner_space", "fasta_mixed_shapes", "fasta_multiline_crlf", "fastq", "fastq_at_quality", "fastq_multiline", "fastq_bad_record"])
def test_parallel_file_reader_equals_sequential_reader(form, tmp_path, monkeypatch):
    """files above 1 MB are mapped, cut at record starts and parsed by all threads (sage2ov_host.cpp::add_plain_file_parallel); multi-line
    FASTQ and anything that does not parse strictly fall back to the sequential reader -- same read set, same counters, either way"""
    rng = np.random.default_rng(3)
    n, L = 40000, 100
    seqs = ["".join(rng.choice(list("ACGTacgtN"), p=[.2475, .2475, .2475, .2475, .00225, .00225, .00225, .00225, .001], size=L - int(rng.integers(0, 30)))) for _ in range(n)]
    path = str(tmp_path / ("x.fq" if form.startswith("fastq") else "x.fa"))
    with open(path, "w", newline="") as f:
        for i, s in enumerate(seqs):
            if form == "fasta":
                f.write(f">r{i}\n{s}\n")
            elif form == "fasta_crlf":
                f.write(f">r{i}\r\n{s}\r\n")
            elif form == "fasta_inner_space":         # white space inside a sequence line is dropped by the reader (every 97th record; a tab in every 389th)
                f.write(f">r{i}\n{s[:11] + ' ' + s[11:] if i % 97 == 5 else (s[:60] + chr(9) + s[60:] + ' ' if i % 389 == 7 else s)}\n")
            elif form == "fasta_mixed_shapes":        # one-line records with a two-line record, an empty record and a '>' inside a header now and then
                if i % 1013 == 3:
                    f.write(f">r{i} >x\n{s[:40]}\n{s[40:]}\n")
                elif i % 1999 == 4:
                    f.write(f">r{i}\n\n")
                else:
                    f.write(f">r{i}\n{s}\n")
            elif form == "fasta_multiline_crlf":
                f.write(f">r{i} x\r\n{s[:37]}\r\n{s[37:]}\r\n")
            elif form == "fastq":
                f.write(f"@r{i}\n{s}\n+\n{'I' * len(s)}\n")
            elif form == "fastq_at_quality":          # quality lines that start with '@' (and '+' lines that repeat the name)
                f.write(f"@r{i}\n{s}\n+r{i}\n@{'+' * (len(s) - 1)}\n")
            elif form == "fastq_multiline":
                f.write(f"@r{i}\n{s[:50]}\n{s[50:]}\n+\n{'I' * 50}\n{'I' * (len(s) - 50)}\n")
            else:                                      # one record in the middle with a short quality line: not strict four-line
                q = "I" * (len(s) - (3 if i == n // 2 else 0))
                f.write(f"@r{i}\n{s}\n+\n{q}\n")
    assert os.path.getsize(path) > (1 << 20)
    (pa, la, fa), sa = _export(path, 21, False, monkeypatch)
    (pb, lb, fb), sb = _export(path, 21, True, monkeypatch)
    assert sa == sb and sa[1] > 0.5 * n
    assert np.array_equal(pa, pb) and np.array_equal(la, lb) and np.array_equal(fa, fb)


@pytest.mark.parametrize("extra1,extra2", [(0, 0), (1, 0), (2, 0), (0, 3)])
def test_parallel_reader_of_two_mate_files_equals_alternating_reader(extra1, extra2, tmp_path, monkeypatch):
    """mate files are read alternately until the file whose turn it is runs out (inputReader.cpp:26-49); with n1 = n2 or n2 + 1 records that is everything, and
    both files go through the parallel reader (add_mate_files_parallel) -- any other pair of counts stays with the sequential reader, which drops the surplus.
    Same read set and counters either way."""
    rng = np.random.default_rng(11)
    n, L = 12000, 100
    def write(path, cnt, tag):
        with open(path, "w") as f:
            for i in range(cnt):
                f.write(f">{tag}{i}\n{''.join(rng.choice(list('ACGT'), size=L - int(rng.integers(0, 20))))}\n")
    p1, p2 = str(tmp_path / "a_1.fa"), str(tmp_path / "a_2.fa")
    write(p1, n + extra1, "a"); write(p2, n + extra2, "b")
    assert os.path.getsize(p1) > (1 << 20) and os.path.getsize(p2) > (1 << 20)
    def load(sequential):
        if sequential:
            monkeypatch.setenv("SAGE2OV_SEQUENTIAL_READER", "1")
        else:
            monkeypatch.delenv("SAGE2OV_SEQUENTIAL_READER", raising=False)
        c = s2.Context(21, device=-2); c.reads_add_file(p1, p2); c.reads_organize()
        st = c.reads_stats(); out = c.reads_export(); c.close()
        return out, (st.total_reads, st.good_reads, st.unique_reads, st.total_bp)
    (pa, la, fa), sa = load(False)
    (pb, lb, fb), sb = load(True)
    expect = 2 * n + (1 if extra1 == 1 else 0) if extra2 == 0 and extra1 <= 1 else (2 * n + 1 if extra1 >= 1 else 2 * n)
    assert sa == sb and sa[0] == expect
    assert np.array_equal(pa, pb) and np.array_equal(la, lb) and np.array_equal(fa, fb)


@pytest.mark.parametrize("form", ["as_written", "lower_case_and_n", "crlf", "spaces", "short_sequence", "wrong_count"])
def test_parallel_reads_file_loader_equals_fscanf_loader(form, tmp_path, monkeypatch):
    """P.reads (readLoader.cpp:289-307) is mapped and parsed by all threads when it is strictly of the shape the writers produce (reads_load_parallel);
    any other shape -- CRLF, blanks as separators, a sequence shorter than its length field, a line count that differs from the header -- goes to the
    fscanf reader with the reference's token semantics.  Same store either way."""
    rng = np.random.default_rng(5)
    n = 3000
    recs = []
    for i in range(n):
        L = int(rng.integers(41, 151)); sq = "".join(rng.choice(list("ACGT"), size=L))
        if form == "lower_case_and_n" and i % 7 == 3:
            sq = sq[:9].lower() + "N" + sq[10:]
        recs.append((int(rng.integers(1, 70000)), L, sq))
    path = str(tmp_path / "t.reads"); sep, nl = ("  " if form == "spaces" else "\t"), ("\r\n" if form == "crlf" else "\n")
    with open(path, "w", newline="") as f:
        f.write(f"{n - 1 if form == 'wrong_count' else n}{nl}")
        for i, (fr, L, sq) in enumerate(recs):
            body = sq[:-5] if form == "short_sequence" and i == n // 2 else sq
            f.write(f"{fr}{sep}{L}{sep}{body}{sep}{body[::-1]}{nl}")
    def load(sequential):
        if sequential:
            monkeypatch.setenv("SAGE2OV_SEQUENTIAL_READER", "1")
        else:
            monkeypatch.delenv("SAGE2OV_SEQUENTIAL_READER", raising=False)
        c = s2.Context(21, device=-2); c.reads_load(path); out = c.reads_export(); nu = c.reads_stats().unique_reads; c.close()
        return out, nu
    (pa, la, fa), na = load(False)
    (pb, lb, fb), nb = load(True)
    assert na == nb == (n - 1 if form == "wrong_count" else n)
    assert np.array_equal(pa, pb) and np.array_equal(la, lb) and np.array_equal(fa, fb)
    assert [int(x) for x in la[1:4]] == [r[1] for r in recs[:3]] and [int(x) for x in fa[1:4]] == [r[0] & 0xFFFF for r in recs[:3]]


def test_step4_needs_the_gpu():
    """no CPU fallback: a context without a device refuses steps 2-4 (here: step 4 and the edge import) with SAGE2OV_ERR_DEVICE"""
    ctx = s2.Context(21, device=-2)
    bases = np.frombuffer(b"ACGTACGTTGCAAGCTAGCTAGGATCCATGCA" * 2, dtype=np.uint8).copy(); off = np.array([0, 32, 64], dtype=np.uint64)
    ctx.reads_add_ascii(bases, off); ctx.reads_organize()
    for call in (ctx.graph_simplify, lambda: ctx.edges_import(np.zeros(0, dtype=s2.EDGE_DTYPE))):
        with pytest.raises(s2.Sage2ovError) as ei:
            call()
        assert ei.value.code == -3
    ctx.close()
