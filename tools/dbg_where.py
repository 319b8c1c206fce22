import sys, os, json, ctypes as C
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, R+'/tests')
os.environ["SAGE2OV_DBG_WHERE"] = "1"
import fixtures as fx, sage2_amd as s2, numpy as np
m = json.loads(sys.argv[1])
bases, off = fx.make_reads(m["synth"])
ctx = s2.Context(m["k"]); ctx.reads_add_ascii(bases, off); ctx.reads_organize()
N = ctx.reads_stats().unique_reads
keys = np.zeros(8 * N, dtype=np.uint64)
assert s2.lib().sage2ov_debug_keys(ctx._h, C.c_void_p(keys.ctypes.data)) == 0
keys = keys.reshape(-1, 2)
ctx.index_build(); st = ctx.index_stats(); print("keys", st.keys, "slots", st.slots)
where = np.zeros(4 * N, dtype=np.uint32)
assert s2.lib().sage2ov_debug_where(ctx._h, C.c_void_p(where.ctypes.data)) == 0
kv = keys[:, 0].astype(object) * (1 << 64) + keys[:, 1].astype(object)
order = np.lexsort((keys[:, 1], keys[:, 0]))
ks = keys[order]; ws = where[order]
same = (ks[1:, 0] == ks[:-1, 0]) & (ks[1:, 1] == ks[:-1, 1])
viol = np.nonzero(same & (ws[1:] != ws[:-1]))[0]
print("distinct keys", 1 + int((~same).sum()), "pairs of equal keys in different slots:", len(viol))
T = st.slots
for v in viol[:12]:
    e1, e2 = int(order[v]), int(order[v + 1])
    print(f"  key {int(ks[v,0]):x}:{int(ks[v,1]):016x} entries e={e1} (read {e1//4+1} t{e1%4}, block {e1//256}, slot {ws[v]}) and e={e2} (read {e2//4+1} t{e2%4}, block {e2//256}, slot {ws[v+1]})  slot diff {(int(ws[v+1])-int(ws[v]))%T}")

ck = np.zeros(16 * N, dtype=np.uint64)
assert s2.lib().sage2ov_debug_countkeys(ctx._h, C.c_void_p(ck.ctypes.data)) == 0
ck = ck.reshape(-1, 4)
bad = np.nonzero((ck[:, 0] != keys[:, 0]) | (ck[:, 1] != keys[:, 1]))[0]
print("entries whose key inside k_index_count differs from k_debug_keys:", len(bad))
for e in bad[:10]:
    print(f"  e={e} read {e//4+1} t{e%4} block {e//256} lane {e%64}: count-kernel {int(ck[e,0]):x}:{int(ck[e,1]):016x}  debug-kernel {int(keys[e,0]):x}:{int(keys[e,1]):016x}")
if len(bad):
    blocks = sorted(set(int(e)//256 for e in bad)); print("blocks", blocks[:40], "types", sorted(set(int(e)%4 for e in bad)))
words, _ = ctx.reads_export_words(); S = ctx.reads_stats().words_per_read
for e in bad[:3]:
    i = e//4+1; print("   words of read", i, [hex(int(x)) for x in words[i*S:(i+1)*S]])
