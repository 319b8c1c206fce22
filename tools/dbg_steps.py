import sys, os
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, R+'/tests')
import fixtures as fx, sage2_amd as s2, numpy as np
name = sys.argv[1] if len(sys.argv) > 1 else "g4_highcopy_k21"
m = fx.golden(name)
bases, off = fx.make_reads(m["synth"])
ctx = s2.Context(m["k"])
ctx.reads_add_ascii(bases, off); ctx.reads_organize(); print("organized", ctx.reads_stats().unique_reads, flush=True)
ctx.index_build(); st = ctx.index_stats(); print("index", st.slots, st.keys, st.csr_entries, st.long_buckets, st.rebuilds, flush=True)
ctx.overlap_initial(); o = ctx.overlap_stats(); print("initial", o.verified_overlaps, o.contained_extension, o.contained_size, flush=True)
ctx.overlap_reduce(); o = ctx.overlap_stats(); print("reduce", o.unresolved_hits, o.edges_inserted, o.transitive_removed, flush=True)
ctx.overlap_convert(); o = ctx.overlap_stats(); print("convert", o.edges, flush=True)
print(m["counters"])
