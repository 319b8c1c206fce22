import sys, os, json, ctypes as C
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, R+'/tests')
import fixtures as fx, sage2_amd as s2, numpy as np
m = json.loads(sys.argv[1])
bases, off = fx.make_reads(m["synth"])
ctx = s2.Context(m["k"]); ctx.reads_add_ascii(bases, off); ctx.reads_organize()
N = ctx.reads_stats().unique_reads
for rep in range(5):
    ctx.index_build(); st = ctx.index_stats()
    out = (C.c_uint64 * 5)()
    import torch; 
    assert s2.lib().sage2ov_debug_table(ctx._h, out) == 0
    occ, inl, unfilled, zerotag, shortcsr = [int(x) for x in out]
    print(f"rep {rep}: 4N={4*N} keys={st.keys} csr={st.csr_entries} long={st.long_buckets} | occupied={occ} inline={inl} unfilled={unfilled} zerotag={zerotag} shortcsr={shortcsr} inline+csr={inl+st.csr_entries}")
