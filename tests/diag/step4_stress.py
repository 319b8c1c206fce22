"""diagnostic: many random synthetic graphs (tests/graphgen.py with varied shapes) through the device step 4 against the step-4 oracle"""
import ctypes, os, sys, tempfile
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, graphgen as gg, sage2_amd as s2
lib = ctypes.CDLL(os.path.join(R, "oracle", "liboracle_step4.so"))
lib.orc4_run_files.argtypes = [ctypes.c_char_p, ctypes.c_ulonglong, ctypes.c_char_p, ctypes.POINTER(ctypes.c_ulonglong)]
n0, n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 1000, int(sys.argv[2]) if len(sys.argv) > 2 else 1400
tmp = tempfile.mkdtemp(); bad = 0
# one context with enough reads for every graph: the read set only provides ids and a uniform length
NMAX = 6000; rng = np.random.default_rng(1); reads = set()
while len(reads) < NMAX: reads.add("".join(rng.choice(list("ACGT"), size=60)))
bases = np.frombuffer("".join(sorted(reads)).encode(), dtype=np.uint8).copy(); off = np.arange(0, (NMAX + 1) * 60, 60, dtype=np.uint64)
ctx = s2.Context(40); ctx.reads_add_ascii(bases, off); ctx.reads_organize()
for seed in range(n0, n1):
    r = np.random.default_rng(seed)
    N, e = gg.random_graph(seed, n_anchor=int(r.integers(1, 60)), n_paths=int(r.integers(1, 150)), max_len=int(r.integers(0, 25)), n_cycles=int(r.integers(0, 8)),
                           p_bad=float(r.choice([0, 0.02, 0.1])), len_hi=int(r.choice([3, 12, 25, 80])))
    if N > NMAX: continue
    ctx.edges_import(e)
    g3, g4, r4 = (os.path.join(tmp, x) for x in ("t.graph3", "t.graph4", "r.graph4"))
    ctx.graph_save(g3); ctx.graph_simplify(); ctx.graph4_save(g4); st = ctx.simplify_stats()
    c = (ctypes.c_ulonglong * 5)(); assert lib.orc4_run_files(g3.encode(), NMAX, r4.encode(), c) == 0
    ok = open(g4, "rb").read() == open(r4, "rb").read() and (st.nodes_contracted, st.removed, st.loop_iterations) == (c[2], c[3], c[1])
    if not ok: bad += 1; print("MISMATCH seed", seed, "N", N, "edges", len(e), (st.nodes_contracted, st.removed, st.loop_iterations), (c[2], c[3], c[1]), flush=True)
print("seeds", n0, "..", n1, "mismatches", bad, flush=True)
ctx.close()
