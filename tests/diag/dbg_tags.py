"""diagnostic: per-read connection counts against the oracle with shrunken tags, in several probe modes"""
import os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import fixtures as fx, oracle_lib as ol, sage2_amd as s2
bits = sys.argv[1] if len(sys.argv) > 1 else "10"
os.environ["SAGE2OV_TEST_TAG_BITS"] = bits
pd = dict(seed=41, genome_len=120000, n_reads=40000, read_len=150, err_ppm=500)
bases, off = fx.make_reads(pd)
o = ol.Oracle(40, threads=8); o.add_reads_ascii(bases, off); o.organize(); o.run_all()
orr, orl, ors, orc = o.export_initial()
for mode in ("default", "SAGE2OV_NO_MINIMIZER_INDEX", "SAGE2OV_SEQUENTIAL_PROBE"):
    for k in ("SAGE2OV_NO_MINIMIZER_INDEX", "SAGE2OV_SEQUENTIAL_PROBE"):
        os.environ.pop(k, None)
    if mode != "default":
        os.environ[mode] = "1"
    g = s2.Context(40, device=0); g.reads_add_ascii(bases, off); g.reads_organize(); g.run_steps23()
    gr, gl, gs, gc = g.overlap_export_initial()
    bad = np.nonzero(gc != orc)[0]
    print(mode, "keys", g.index_stats().keys, "oracle keys", o.counter("keys"), "conn mismatches", len(bad), "first", bad[:5], gc[bad[:5]], orc[bad[:5]],
          "right mism", int((gr[1:] != orr[1:]).sum()), "slow reads", g.timings().sequential_reads)
    g.close()
