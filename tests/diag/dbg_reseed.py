"""diagnostic: the reseed path of the index build (impure long buckets) with shrunken tags on the high-copy fixture"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import fixtures as fx, sage2_amd as s2
for bits in sys.argv[1:]:
    os.environ["SAGE2OV_TEST_TAG_BITS"] = bits
    m = fx.golden("g4_highcopy_k21"); bases, off = fx.make_reads(m["synth"])
    g = s2.Context(m["k"], device=0); g.reads_add_ascii(bases, off); g.reads_organize()
    try:
        g.run_steps23()
        st = g.index_stats()
        import tempfile
        p = tempfile.mktemp(); g.graph_save(p)
        same = open(p, "rb").read() == fx.golden_graph3("g4_highcopy_k21")
        print("bits", bits, "rebuilds", st.rebuilds, "long", st.long_buckets, "want long", m["counters"]["long_buckets"], "graph3 identical", same)
    except Exception as e:
        print("bits", bits, "error:", e)
    g.close()
