"""diagnostic: step 4 on the graph of a few million noisy reads (with repeats), device against the step-4 oracle (md5 of P.graph4)"""
import ctypes, hashlib, os, sys, tempfile, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import fixtures as fx, sage2_amd as s2
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
lib = ctypes.CDLL(os.path.join(R, "oracle", "liboracle_step4.so")); lib.orc4_run_files.argtypes = [ctypes.c_char_p, ctypes.c_ulonglong, ctypes.c_char_p, ctypes.POINTER(ctypes.c_ulonglong)]
tmp = tempfile.mkdtemp()
pd = dict(seed=77, genome_len=3 * n, n_reads=n, read_len=150, err_ppm=1500, n_repeat_families=6, repeat_copies=40, repeat_len=300)
bases, off = fx.make_reads(pd)
ctx = s2.Context(40); ctx.reads_add_ascii(bases, off); ctx.reads_organize(); ctx.run_steps23()
g3, g4, r4 = (os.path.join(tmp, x) for x in ("t.graph3", "t.graph4", "r.graph4"))
ctx.graph_save(g3); ctx.graph_simplify(); st = ctx.simplify_stats(); ctx.graph4_save(g4); N = ctx.reads_stats().unique_reads; ctx.close()
t0 = time.time(); c = (ctypes.c_ulonglong * 5)(); assert lib.orc4_run_files(g3.encode(), N, r4.encode(), c) == 0; t1 = time.time()
md5 = lambda p: hashlib.md5(open(p, "rb").read()).hexdigest()
print("reads", n, "unique", N, "device", (st.nodes_contracted, st.removed, st.loop_iterations), "%.1f ms" % st.device_ms, "oracle", (c[2], c[3], c[1]), "%.1f s incl. text parsing" % (t1 - t0),
      "graph4 identical", md5(g4) == md5(r4), flush=True)
