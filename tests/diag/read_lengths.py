"""diagnostic: step time per read for several read lengths, clean and with 0.1 % errors.  usage: read_lengths.py [n_reads] [L ...]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import fixtures as fx, sage2_amd as s2
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
for L in ([int(x) for x in sys.argv[2:]] or (100, 150, 200, 250, 300, 400, 500)):
    for err in (0, 1000):
        pd = dict(seed=5, genome_len=n * L // 50, n_reads=n, read_len=L, err_ppm=err)
        bases, off = fx.make_reads(pd)
        ctx = s2.Context(40); ctx.reads_add_ascii(bases, off); ctx.reads_organize(); ctx.run_steps23()
        t0 = time.time(); ctx.run_steps23(); t1 = time.time(); tm = ctx.timings(); st = ctx.overlap_stats()
        print(f"L {L} err {err}: {1e3 * (t1 - t0):.1f} ms ({1e9 * (t1 - t0) / ctx.reads_stats().unique_reads:.1f} ns/read) index {tm.index_ms:.1f} probe {tm.probe_ms:.1f} (kernel {tm.probe_kernel_ms:.1f}, sequential reads {tm.sequential_reads}) "
              f"reduce {tm.reduce_ms:.1f}; overlaps {st.verified_overlaps}, unresolved {st.left_to_explore}", flush=True)
        ctx.close()
