#!/usr/bin/env python3
"""oracle/make_golden_step4.py -- TEST INFRASTRUCTURE: post-step-4 graphs of the golden fixtures, made by the REFERENCE's own classes.

For every fixture of oracle/make_golden.py: the FASTA is regenerated, `SAGE2 -M 3 -s` (oracle/_ref/SAGE2) writes t.reads / t.graph3, and
`sage2ref_run_step4` (oracle/ref_driver.cpp: the reference's loaders, its simplification loop in main.cpp:150-172 order, its writer)
dumps the in-memory graph after step 4 -- a file the reference itself never writes although its step 5 would read it as `<prefix>.graph4`.
Stored: tests/golden/<name>.graph4.gz (verbatim) and tests/golden/<name>.step4.json (md5, size, counters).  Build container only.
"""
import ctypes, gzip, json, os, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import make_golden as mg

def run_step4(drv, prefix, k, out, threads=8):
    t = (ctypes.c_double * 2)(); c = (ctypes.c_ulonglong * 4)()
    drv.sage2ref_run_step4.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_ulonglong)]
    assert drv.sage2ref_run_step4(prefix.encode(), k, threads, out.encode(), t, c) == 0
    return dict(unique_reads=c[0], loop_iterations=c[1], nodes_contracted=c[2], removed=c[3]), list(t)

def main():
    lib = ctypes.CDLL(os.path.join(ROOT, "sage2_amd", "libsage2ov.so"))
    lib.sage2ov_synth_write_fasta.argtypes = [ctypes.POINTER(mg.SynthParams), ctypes.c_char_p]
    drv = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libsage2ref_driver.so"))
    gold = os.path.join(ROOT, "tests", "golden")
    only = set(sys.argv[1:])
    for name, (k, threads, pd) in mg.FIXTURES.items():
        if only and name not in only: continue
        tmp = tempfile.mkdtemp(prefix="sage2gold4_")
        try:
            fa = os.path.join(tmp, "x.fa")
            if "recipe" in pd:
                sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
                import fixtures
                fixtures.write_recipe_fasta(pd, fa)
            else:
                p = mg.SynthParams(**pd)
                assert lib.sage2ov_synth_write_fasta(ctypes.byref(p), fa.encode()) == 0
            env = dict(os.environ, OMP_NUM_THREADS=str(threads), LC_ALL="C")
            subprocess.run([mg.REF, "-f", fa, "-k", str(k), "-o", os.path.join(tmp, "out"), "-p", "t", "-M", "3", "-s"], check=True, env=env, stdout=subprocess.DEVNULL)
            meta3 = json.load(open(os.path.join(gold, name + ".json")))
            assert mg.md5(os.path.join(tmp, "out", "t.graph3")) == meta3["graph3_md5"], "graph3 differs from the committed fixture"
            g4 = os.path.join(tmp, "out", "t.graph4")
            cnt, t = run_step4(drv, os.path.join(tmp, "out", "t"), k, g4, threads)
            meta = dict(name=name, k=k, graph4_md5=mg.md5(g4), graph4_size=os.path.getsize(g4), counters=cnt)
            with open(g4, "rb") as fi, gzip.GzipFile(os.path.join(gold, name + ".graph4.gz"), "wb", mtime=0) as fo:
                shutil.copyfileobj(fi, fo)
            json.dump(meta, open(os.path.join(gold, name + ".step4.json"), "w"), indent=1, sort_keys=True)
            print(name, cnt, "graph4", meta["graph4_size"], "gz", os.path.getsize(os.path.join(gold, name + ".graph4.gz")), "t", [round(x, 3) for x in t])
        finally:
            shutil.rmtree(tmp, ignore_errors=True)

if __name__ == "__main__":
    main()
