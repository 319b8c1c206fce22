"""diagnostic: step-4 statistics on the synthetic graphs and device time on a C2-like read set"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, fixtures as fx, sage2_amd as s2, graphgen as gg
for seed in (0, 5, 13, 27, 39):
    N, e = gg.random_graph(seed, n_anchor=8 + 3 * seed, n_paths=20 + 6 * seed, max_len=3 + seed % 9, n_cycles=seed % 4, p_bad=0.02 * (seed % 3))
    rng = np.random.default_rng(seed); reads = set()
    while len(reads) < N: reads.add("".join(rng.choice(list("ACGT"), size=100)))
    bases = np.frombuffer("".join(sorted(reads)).encode(), dtype=np.uint8).copy(); off = np.arange(0, (N + 1) * 100, 100, dtype=np.uint64)
    ctx = s2.Context(40); ctx.reads_add_ascii(bases, off); ctx.reads_organize(); ctx.edges_import(e); ctx.graph_simplify(); st = ctx.simplify_stats()
    print("seed", seed, "N", N, "edges", len(e), "contracted", st.nodes_contracted, "removed", st.removed, "iters", st.loop_iterations, "left", st.edges, "on edges", st.reads_on_edges, "ms %.2f" % st.device_ms, flush=True)
    ctx.close()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
for err in (0, 1000):
    pd = dict(seed=2, genome_len=3 * n, n_reads=n, read_len=150, err_ppm=err)
    bases, off = fx.make_reads(pd)
    ctx = s2.Context(40); ctx.reads_add_ascii(bases, off); ctx.reads_organize(); ctx.run_steps23()
    for rep in range(2):
        t0 = time.time(); ctx.graph_simplify(); t1 = time.time(); st = ctx.simplify_stats()
        print("reads", n, "err", err, "edges in", ctx.overlap_stats().edges, "contracted", st.nodes_contracted, "removed", st.removed, "iters", st.loop_iterations, "left", st.edges,
              "on edges", st.reads_on_edges, "device ms %.1f" % st.device_ms, "wall %.3f s" % (t1 - t0), flush=True)
    t0 = time.time(); ctx.graph4_save("/tmp/t.graph4"); print("graph4 written in %.2f s, %.0f MB" % (time.time() - t0, os.path.getsize("/tmp/t.graph4") / 1e6), flush=True)
    ctx.close()
