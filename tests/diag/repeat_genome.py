"""diagnostic: a genome with high-copy repeats (long buckets -> the reduce phase takes the exact serial replay on the host): phase times"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import fixtures as fx, sage2_amd as s2
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
cases = ((0, 4, 300), (1000, 4, 300), (1000, 0, 0)) if len(sys.argv) < 3 else ((1000, int(sys.argv[2]), 300),)
for err, fam, copies in cases:
    pd = dict(seed=3, genome_len=3 * n, n_reads=n, read_len=150, err_ppm=err, n_repeat_families=fam, repeat_copies=copies, repeat_len=400)
    bases, off = fx.make_reads(pd)
    ctx = s2.Context(40); ctx.reads_add_ascii(bases, off); ctx.reads_organize()
    t0 = time.time(); ctx.run_steps23(); t1 = time.time()
    tm = ctx.timings(); st = ctx.overlap_stats(); ix = ctx.index_stats()
    print(f"reads {n} err {err} repeats {fam}x{copies}: long buckets {ix.long_buckets}, unresolved {st.left_to_explore}, edges {st.edges}; total {1e3 * (t1 - t0):.0f} ms: index {tm.index_ms:.1f} probe {tm.probe_ms:.1f} "
          f"reciprocal {tm.reciprocal_ms:.1f} reduce {tm.reduce_ms:.1f} convert {tm.convert_ms:.1f}", flush=True)
    ctx.close()
