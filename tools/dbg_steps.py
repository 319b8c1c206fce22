import sys, os, json, time, faulthandler
faulthandler.enable()
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, R+'/tests')
import fixtures as fx, sage2_amd as s2, numpy as np
arg = sys.argv[1] if len(sys.argv) > 1 else "g4_highcopy_k21"
m = json.loads(arg) if arg.startswith("{") else fx.golden(arg)
p = fx.synth_params(m["synth"]); g = s2.synth_genome(p)
ctx = s2.Context(m["k"])
t=time.time(); ctx.reads_add_synth(p, g); ctx.reads_organize(); print("organized", ctx.reads_stats().unique_reads, f"{time.time()-t:.1f}s", flush=True)
t=time.time(); ctx.index_build(); st = ctx.index_stats(); print("index", st.slots, st.keys, st.csr_entries, st.long_buckets, st.rebuilds, f"{time.time()-t:.3f}s", flush=True)
t=time.time(); ctx.overlap_initial(); o = ctx.overlap_stats(); print("initial", o.verified_overlaps, o.contained_extension, o.contained_size, f"{time.time()-t:.3f}s", flush=True)
t=time.time(); ctx.overlap_reduce(); o = ctx.overlap_stats(); print("reduce", o.unresolved_hits, o.edges_inserted, o.transitive_removed, f"{time.time()-t:.3f}s", flush=True)
t=time.time(); ctx.overlap_convert(); o = ctx.overlap_stats(); print("convert", o.edges, f"{time.time()-t:.3f}s", flush=True)
tm = ctx.timings(); print({k: getattr(tm, k) for k, _ in tm._fields_})
