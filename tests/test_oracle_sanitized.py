"""CPU: the oracle restatement rebuilt with -fsanitize=address,undefined (oracle/Makefile: liboracle_asan.so) replays golden fixtures in a
child interpreter (libasan preloaded); any out-of-bounds access, use-after-free or undefined behaviour aborts the child.  The files it
writes must still be the reference's."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import oracle_lib as ol
ol.ORACLE_SO = os.path.join(sys.argv[1], "oracle", "liboracle_asan.so")
import fixtures as fx
for name in sys.argv[3:]:
    m = fx.golden(name)
    bases, off = fx.make_reads(m["synth"])
    o = ol.Oracle(m["k"], threads=2); o.add_reads_ascii(bases, off); o.organize(); o.run_all()
    gp = os.path.join(sys.argv[2], name + ".graph3"); rp = os.path.join(sys.argv[2], name + ".reads")
    o.write_graph3(gp); o.write_reads(rp)
    assert fx.graph3_matches(gp, name) and fx.md5_file(rp) == m["reads_md5"], name
    o.close()
print("SAN_OK")
'''


def test_oracle_under_asan_ubsan(tmp_path):
    so = os.path.join(ROOT, "oracle", "liboracle_asan.so")
    if not os.path.exists(so):
        r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "liboracle_asan.so"], capture_output=True, text=True)
        if r.returncode != 0:
            pytest.skip("sanitizer build not available: " + r.stderr[-300:])
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan):
        pytest.skip("libasan not found")
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", OMP_NUM_THREADS="2")
    # mixed lengths (containment), long buckets (serial BFS order), k > 64 (two-word keys), palindromes / tandem repeats
    names = ["g5_mixedlen_k21", "g4_highcopy_k21", "g6_k70_150", "g7_palindrome_tandem_k21"]
    r = subprocess.run([sys.executable, "-c", CHILD, ROOT, str(tmp_path)] + names, capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0 and "SAN_OK" in r.stdout, (r.stdout[-2000:] + r.stderr[-4000:])
