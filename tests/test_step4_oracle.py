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
