"""The step-4 restatement (oracle/step4_oracle.cpp) against graphs dumped by the reference's own classes (tests/golden/*.graph4.gz,
made by oracle/make_golden_step4.py): byte identity of the whole file, i.e. the surviving edges, their read lists and their order."""
import ctypes, gzip, hashlib, json, os
import pytest
import fixtures as fx

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lib():
    p = os.path.join(ROOT, "oracle", "liboracle_step4.so")
    if not os.path.exists(p):
        import subprocess
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "liboracle_step4.so"], check=True, stdout=subprocess.DEVNULL)
    lib = ctypes.CDLL(p)
    lib.orc4_run_files.argtypes = [ctypes.c_char_p, ctypes.c_ulonglong, ctypes.c_char_p, ctypes.POINTER(ctypes.c_ulonglong)]
    return lib


@pytest.mark.parametrize("name", fx.golden_names())
def test_step4_restatement_equals_reference_dump(name, tmp_path):
    meta = json.load(open(os.path.join(ROOT, "tests", "golden", name + ".step4.json")))
    g3 = tmp_path / "t.graph3"; g3.write_bytes(fx.golden_graph3(name))
    out = tmp_path / "t.graph4"
    c = (ctypes.c_ulonglong * 5)()
    assert _lib().orc4_run_files(str(g3).encode(), meta["counters"]["unique_reads"], str(out).encode(), c) == 0
    want = gzip.open(os.path.join(ROOT, "tests", "golden", name + ".graph4.gz")).read()
    got = out.read_bytes()
    assert hashlib.md5(want).hexdigest() == meta["graph4_md5"]
    assert c[1] == meta["counters"]["loop_iterations"] and c[2] == meta["counters"]["nodes_contracted"] and c[3] == meta["counters"]["removed"]
    assert got == want


def _ref_driver():
    p = os.path.join(ROOT, "oracle", "_ref", "libsage2ref_driver.so")
    if not os.path.exists(p):
        pytest.skip("oracle/_ref not built (the reference is only present in the build container)")
    drv = ctypes.CDLL(p)
    if not hasattr(drv, "sage2ref_run_step4"):
        pytest.skip("oracle/_ref predates the step-4 driver")
    drv.sage2ref_run_step4.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_ulonglong)]
    return drv


@pytest.mark.parametrize("seed", range(24))
def test_step4_restatement_equals_reference_on_synthetic_graphs(seed, tmp_path):
    """cycles, closed and parallel chains, tips, multi-edges, nodes that do not combine -- ids scattered (tests/graphgen.py); the reference's
    classes run in process through oracle/_ref (build container only)"""
    import graphgen as gg
    drv = _ref_driver()
    N, e = gg.random_graph(seed, n_anchor=8 + 3 * seed, n_paths=20 + 6 * seed, max_len=3 + seed % 9, n_cycles=seed % 4, p_bad=0.02 * (seed % 3))
    pre = str(tmp_path / "t"); gg.write_graph3(pre + ".graph3", N, e); gg.write_reads(pre + ".reads", N, seed=seed)
    t = (ctypes.c_double * 2)(); c = (ctypes.c_ulonglong * 4)()
    assert drv.sage2ref_run_step4(pre.encode(), 40, 1, (pre + ".ref4").encode(), t, c) == 0
    c2 = (ctypes.c_ulonglong * 5)()
    assert _lib().orc4_run_files((pre + ".graph3").encode(), N, (pre + ".orc4").encode(), c2) == 0
    assert (c[1], c[2], c[3]) == (c2[1], c2[2], c2[3])
    assert open(pre + ".orc4", "rb").read() == open(pre + ".ref4", "rb").read()
