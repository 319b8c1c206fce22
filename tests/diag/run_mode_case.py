"""diagnostic: for the reads whose records differ between run mode and the general path (tests/diag/run_mode_diff.py), the buckets of the windows involved"""
import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import fixtures as fx, sage2_amd as s2
n = int(sys.argv[1]); k = 40; L = 150
p = fx.synth_params(dict(seed=3 if n == 50_000_000 else 2, genome_len=3 * n, n_reads=n, read_len=L))
def ctx_run(norun):
    if norun: os.environ["SAGE2OV_NO_RUN_MODE"] = "1"
    else: os.environ.pop("SAGE2OV_NO_RUN_MODE", None)
    ctx = s2.Context(k, device=0); ctx.reads_add_synth(p, s2.synth_genome(p)); ctx.reads_organize()
    ctx.index_build(); ctx.overlap_initial()
    return ctx
a = ctx_run(False); ra, la, sa, ca = a.overlap_export_initial()
b = ctx_run(True); rb, lb, sb, cb = b.overlap_export_initial()
packed, length, freq = b.reads_export()
def bases(i): 
    bits = np.unpackbits(packed[i]); return "".join("ACGT"[2 * int(bits[2 * x]) + int(bits[2 * x + 1])] for x in range(L))
def rc(s): return s[::-1].translate(str.maketrans("ACGT", "TGCA"))
def key(s):          # (v0 = leading h - 32 bases, v1 = last 32 bases), right aligned
    v = 0
    for ch in s: v = v * 4 + "ACGT".index(ch)
    return v >> 64, v & ((1 << 64) - 1)
d = np.nonzero(ca != cb)[0]
for i in d[:4]:
    i = int(i); R = bases(i)
    print("read", i, "conn run", int(ca[i]), "general", int(cb[i]))
    tot = 0
    for j in range(L - k + 1):
        ents, cnt = b.index_lookup(*key(R[j:j + k]))
        for e in ents:
            r2, t = e >> 2, e & 3
            if r2 == i: continue
            X = bases(r2)
            if t == 0: okv = X[:L - j] == R[j:]
            elif t == 2: okv = rc(X)[:L - j] == R[j:]
            elif t == 1: okv = X[L - (j + k):] == R[:j + k]
            else: okv = rc(X)[L - (j + k):] == R[:j + k]
            tot += okv
            if not okv: print("   window", j, "entry", r2, "type", t, "is NOT an overlap (bucket of", cnt, "entries:", [(x >> 2, x & 3) for x in ents], ")")
    print("   true overlaps by the table:", tot)
dr = np.nonzero(ra != rb)[0]
for i in dr[:3]:
    i = int(i); R = bases(i)
    for nm, rec in (("run", ra[i]), ("general", rb[i])):
        x = int(rec & ((1 << 40) - 1)); ty = int((rec >> 40) & 3); ln = int(rec >> 42)
        X = bases(x); Xo = rc(X) if ty else X
        offs = [o for o in range(-L + 1, L) if (Xo[max(0, -o):L - max(0, o)] == R[max(0, o):L - max(0, -o)])]
        print("read", i, nm, "right record: read", x, "orientation", ty, "overhang", ln, "-> offsets at which it really overlaps this read:", offs)
    print("   R =", R)
    print("   windows 0..9 buckets:", [(j, [(e >> 2, e & 3) for e in b.index_lookup(*key(R[j:j + k]))[0]]) for j in range(10)])
