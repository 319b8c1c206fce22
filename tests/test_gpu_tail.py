"""GPU: acceptance through the UNCHANGED reference tail.  Files WRITTEN BY THE GPU PATH IN THIS TEST -- `sage2ov -M 3`
(P.reads + P.graph3) and `sage2ov -M 4` (P.graph4) -- are handed to the reference binary (oracle/_ref/SAGE2, built from /root/reference
by oracle/Makefile; it travels to the GPU box as a built file), which continues with `-m 4 -M 7` resp. `-m 5 -M 7`; the result is compared
with the reference's own one-shot run of all seven steps on the same FASTA (main.cpp:141-148 loader, :246 / :290 contig and scaffold
writers).  Error-free fixtures: contigs and scaffolds, byte for byte.  Noisy fixtures: the graph files the tail writes (P.graph5, P.graph6)
-- the reference does not reproduce its own contigs across ANY restart on noisy data."""
import os
import shutil
import subprocess

import pytest

import fixtures as fx
import sage2_amd as s2

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("minimiser_groups_on")]      # (small inputs: the groups' half of the look-up code is exercised by request, conftest.py)
REF = os.path.join(fx.ROOT, "oracle", "_ref", "SAGE2")
CLI = os.path.join(fx.ROOT, "sage2_amd", "sage2ov")


def _same(a, b):
    return os.path.exists(a) and os.path.exists(b) and open(a, "rb").read() == open(b, "rb").read()


# Not usable here: g3 / g5 / g8 -- the reference's OWN one-shot run dies in its steps 5-7 on them (SIGFPE in the flow / mate-pair code); g4 / g9 --
# high-copy repeats: the min-cost flow has ties that CS2 breaks by arc order, which no restart of the reference reproduces (DESIGN.md
# section 10).  Their hand-over files (P.graph3, P.graph4) are compared byte for byte in test_gpu_parity.py / test_gpu_step4.py.
@pytest.mark.parametrize("name,start", [("g1_clean100_k21", 4), ("g2_clean150_k40", 4), ("g6_k70_150", 4),
                                        ("g1_clean100_k21", 5), ("g2_clean150_k40", 5), ("g6_k70_150", 5)])
def test_reference_tail_on_gpu_written_files(name, start, tmp_path):
    if not os.path.exists(REF):
        pytest.skip("oracle/_ref/SAGE2 not built (it is built in the build container and travels with the snapshot)")
    m = fx.golden(name)
    fa = str(tmp_path / "x.fa"); s2.synth_write_fasta(fx.synth_params(m["synth"]), fa)
    env = dict(os.environ, OMP_NUM_THREADS="8", LC_ALL="C")
    k = str(m["k"])
    # the GPU path writes the hand-over files ...
    ours = str(tmp_path / "ours")
    subprocess.run([CLI, "-f", fa, "-k", k, "-o", ours, "-p", "t", "-M", str(start - 1)], check=True, stdout=subprocess.DEVNULL)
    assert fx.md5_file(os.path.join(ours, "t.reads")) == m["reads_md5"]
    assert os.path.exists(os.path.join(ours, "t.graph%d" % (start - 1)))
    # ... the unchanged reference continues from them ...
    subprocess.run([REF, "-f", fa, "-k", k, "-o", ours, "-p", "t2", "-i", "t", "-m", str(start), "-M", "7", "-s"], check=True, env=env, stdout=subprocess.DEVNULL, cwd=str(tmp_path))
    # ... and must end where its own one-shot run ends
    full = str(tmp_path / "full")
    subprocess.run([REF, "-f", fa, "-k", k, "-o", full, "-p", "t", "-M", "7", "-s"], check=True, env=env, stdout=subprocess.DEVNULL, cwd=str(tmp_path))
    assert _same(os.path.join(ours, "t.graph3"), os.path.join(full, "t.graph3")) or start == 5
    suffixes = [".graph5", ".graph6"] + (["_contig.fasta", "_scaffold.fasta"] if "clean" in name else [])
    for sfx in suffixes:
        assert _same(os.path.join(ours, "t2" + sfx), os.path.join(full, "t" + sfx)), f"{name} from step {start}: {sfx} differs"
    shutil.rmtree(ours, ignore_errors=True); shutil.rmtree(full, ignore_errors=True)


@pytest.mark.parametrize("name", ["g1_clean100_k21", "g3_noisy_rep_k21", "g4_highcopy_k21", "g5_mixedlen_k21", "g6_k70_150"])
def test_in_memory_handover_to_the_references_classes(name, tmp_path):
    """INTEGRATION.md section 2, compiled and run: the canonical edge list the GPU computed (sage2ov_edges_export) is fed to the REFERENCE's own
    OverlapGraph::insertEdgeInGraph in list order (oracle/ref_driver.cpp::sage2ref_step4_from_edges, linked against the reference's objects), the
    reference runs ITS step 4 on that in-memory graph and dumps it with ITS writer: the result must be the graph4 fixture, which the reference
    produced from its own P.graph3 -- i.e. handing the edges over in memory is equivalent to going through the file."""
    import ctypes as C, gzip
    import numpy as np
    drv = os.path.join(fx.ROOT, "oracle", "_ref", "libsage2ref_driver.so")
    if not os.path.exists(drv):
        pytest.skip("oracle/_ref not built")
    L = C.CDLL(drv)
    if not hasattr(L, "sage2ref_step4_from_edges"):
        pytest.skip("oracle/_ref is older than this test: rebuild it in the build container (make -C oracle ref)")
    m = fx.golden(name)
    bases, off = fx.make_reads(m["synth"])
    ctx = s2.Context(m["k"], device=0); ctx.reads_add_ascii(bases, off); ctx.reads_organize(); ctx.run_steps23()
    rp = str(tmp_path / "t.reads"); ctx.reads_save(rp)
    e = ctx.edges(); st = ctx.reads_stats()
    flat = np.stack([e["from"], e["to"], e["type"].astype(np.uint64), e["length"].astype(np.uint64)], axis=1).astype(np.uint64).copy()
    out = str(tmp_path / "t.graph4"); c = (C.c_ulonglong * 4)()
    L.sage2ref_step4_from_edges.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_void_p, C.c_ulonglong, C.c_ulonglong, C.c_ulonglong, C.c_char_p, C.POINTER(C.c_ulonglong)]
    assert L.sage2ref_step4_from_edges(rp.encode(), m["k"], 4, flat.ctypes.data, len(e), st.good_reads, st.average_read_length, out.encode(), c) == 0
    want = gzip.open(os.path.join(fx.GOLDEN, name + ".graph4.gz"), "rb").read()
    assert open(out, "rb").read() == want
    ctx.close()
