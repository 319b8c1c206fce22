import sys, os, json, ctypes as C
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, R+'/tests')
import fixtures as fx, sage2_amd as s2, numpy as np
m = json.loads(sys.argv[1])
bases, off = fx.make_reads(m["synth"])
ctx = s2.Context(m["k"]); ctx.reads_add_ascii(bases, off); ctx.reads_organize()
N = ctx.reads_stats().unique_reads
outs = []
for rep in range(4):
    if rep == 2: ctx.index_build()
    out = np.zeros(8 * N, dtype=np.uint64)
    assert s2.lib().sage2ov_debug_keys(ctx._h, C.c_void_p(out.ctypes.data)) == 0
    outs.append(out)
for rep in range(1, 4):
    d = np.nonzero(outs[rep] != outs[0])[0]
    print("rep", rep, "differing words vs rep0:", len(d), d[:10] // 8 + 1 if len(d) else "")
# host reference from the word image
words, _ = ctx.reads_export_words(); S = ctx.reads_stats().words_per_read
def bits(w, pos, n):   # n bits from bit pos of big-endian word array
    v = 0
    for x in w: v = (v << 64) | int(x)
    tot = 64 * len(w)
    return (v >> (tot - pos - n)) & ((1 << n) - 1)
bad = 0
h = min(m["k"], 64)
for i in list(range(1, 200)) + list(range(N - 5000, N + 1)):
    w = words[i * S:(i + 1) * S]; L = int(w[S - 1]) & 0x1FF
    pre = bits(w, 0, 2 * h); suf = bits(w, 2 * (L - h), 2 * h)
    def rck(v):
        r = 0
        for b in range(h): r = (r << 2) | (3 - ((v >> (2 * b)) & 3))
        return r
    want = [pre, suf, rck(suf), rck(pre)]
    for t in range(4):
        e = (i - 1) * 4 + t
        got = (int(outs[0][2 * e]) << 64) | int(outs[0][2 * e + 1])
        if got != want[t]:
            bad += 1
            if bad < 6: print("key mismatch read", i, "type", t, hex(got), hex(want[t]))
print("host-vs-device key mismatches:", bad)
