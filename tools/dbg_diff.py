import sys, os
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, R+'/tests')
import fixtures as fx, sage2_amd as s2, oracle_lib as ol, numpy as np
import json
name = sys.argv[1] if len(sys.argv) > 1 else "g5_mixedlen_k21"
m = json.loads(name) if name.startswith("{") else fx.golden(name)
bases, off = fx.make_reads(m["synth"])
ctx = s2.Context(m["k"]); ctx.reads_add_ascii(bases, off); ctx.reads_organize(); ctx.run_steps23()
o = ol.Oracle(m["k"], 8); o.add_reads_ascii(bases, off); o.organize(); o.run_all()
gr, gl, gs, gc = ctx.overlap_export_initial(); orr, orl, ors, orc = o.export_initial()
def dec(x): return (int(x) & ((1<<40)-1), (int(x)>>40)&3, int(x)>>42)
for nm, a, b in (("conn", gc, orc), ("right", gr, orr), ("left", gl, orl)):
    d = np.nonzero(a != b)[0]
    print(nm, "mismatches:", len(d))
    for i in d[:8]: print("   id", i, "gpu", dec(a[i]) if nm!="conn" else a[i], "oracle", dec(b[i]) if nm!="conn" else b[i])
cls = lambda s: np.where(np.isin(s, (1, 2)), 0, s)
d = np.nonzero(cls(gs) != cls(ors))[0]; print("status mismatches", len(d), [(int(i), int(gs[i]), int(ors[i])) for i in d[:10]])
e, oe = ctx.edges(), o.export_edges()
print("edges", len(e), len(oe))
ge = set(zip(e["from"].tolist(), e["to"].tolist(), e["type"].tolist(), e["length"].tolist(), e["length_twin"].tolist()))
oo = set(map(tuple, oe.tolist()))
print("only gpu", sorted(ge - oo)[:10]); print("only oracle", sorted(oo - ge)[:10])
st = ctx.overlap_stats(); print("gpu stats", st.verified_overlaps, st.contained_extension, st.contained_size, st.edges_inserted, st.transitive_removed, "oracle", o.counters())
_, ln, _ = o.export_reads()
for (a,b,t,l,lt) in sorted(ge - oo)[:5]: print("  lens", a, ln[a], b, ln[b], "status", gs[a], gs[b], ors[a], ors[b])
