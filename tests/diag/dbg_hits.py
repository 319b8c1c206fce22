import sys, os, json
R=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, R+'/tests')
import fixtures as fx, sage2_amd as s2, oracle_lib as ol, numpy as np
m = json.loads(sys.argv[1])
bases, off = fx.make_reads(m["synth"])
ctx = s2.Context(m["k"]); ctx.reads_add_ascii(bases, off); ctx.reads_organize(); ctx.index_build(); ctx.overlap_initial()
o = ol.Oracle(m["k"], 8); o.add_reads_ascii(bases, off); o.organize(); o.build_index(); o.initial()
gh = ctx.debug_all_hits()
gc = ctx.overlap_export_initial()[3]; oc = o.export_initial()[3]
bad = np.nonzero(gc != oc)[0]
print("conn mismatches", len(bad), "gpu hits rows", len(gh))
starts = np.searchsorted(gh[:,0], np.arange(len(gc)+1))
nshow = 0
for i in bad[:2000]:
    g = gh[starts[i]:starts[i+1]]
    oh = o.debug_hits(i)
    gs = [(int(a), int(b), int(np.int32(c))) for a,b,c in g[:,1:4]]
    os_ = [tuple(int(x) for x in r) for r in oh]
    if gs != os_ and nshow < 6:
        nshow += 1
        miss = [x for x in os_ if x not in gs]; extra = [x for x in gs if x not in os_]
        print("read", i, "mode1 gpu", len(gs), "oracle", len(os_), "missing", miss, "extra", extra, "conn gpu/or", gc[i], oc[i])
        # where in the oracle list are the missing ones
        for x in miss: print("    position in oracle list", os_.index(x), "of", len(os_))
print("reads whose MODE1 list differs:", sum(1 for i in bad[:2000] if [(int(a), int(b), int(np.int32(c))) for a,b,c in gh[starts[i]:starts[i+1]][:,1:4]] != [tuple(int(x) for x in r) for r in o.debug_hits(i)]))
# are the missed reads present in the index?
fwd, ln, _ = o.export_reads()
def seq(i):
    b=fwd[i]; L=int(ln[i]); return ''.join('ACGT'[(b[p>>2]>>(6-2*(p&3)))&3] for p in range(L))
def rc(s): return s[::-1].translate(str.maketrans('ACGT','TGCA'))
def key_of(s):
    v=0
    for ch in s: v=(v<<2)|'ACGT'.index(ch)
    return v>>64, v & ((1<<64)-1)
h = min(m["k"], 64)
missing_entries = 0; checked = 0
hi_ids = list(range(len(ln)-3000, len(ln)))
for r2 in hi_ids:
    s = seq(r2); r = rc(s)
    for t, ks in enumerate((s[:h], s[-h:], r[:h], r[-h:])):
        v0, v1 = key_of(ks)
        ent, n = ctx.index_lookup(v0, v1)
        want, wn = o.lookup(v0, v1)
        checked += 1
        if ent != want[:len(ent)] or n != wn:
            missing_entries += 1
            if missing_entries < 10: print("index mismatch for read", r2, "type", t, "gpu", [(e>>2,e&3) for e in ent], "oracle", [(e>>2,e&3) for e in want])
print("index entries checked", checked, "mismatching", missing_entries)
