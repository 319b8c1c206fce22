"""diagnostic: step time per read at several k (150-bp reads, 2 M reads at 50x): k <= 22 runs the three-windows-per-lane instantiation"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import fixtures as fx, sage2_amd as s2
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
for k in (21, 31, 40, 55, 70, 100):
    for err in (0, 1000):
        pd = dict(seed=5, genome_len=n * 3, n_reads=n, read_len=150, err_ppm=err)
        bases, off = fx.make_reads(pd)
        ctx = s2.Context(k); ctx.reads_add_ascii(bases, off); ctx.reads_organize(); ctx.run_steps23()
        t0 = time.time(); ctx.run_steps23(); t1 = time.time(); tm = ctx.timings(); st = ctx.overlap_stats()
        print(f"k {k} err {err}: {1e3 * (t1 - t0):.1f} ms ({1e9 * (t1 - t0) / ctx.reads_stats().unique_reads:.1f} ns/read) index {tm.index_ms:.1f} probe {tm.probe_ms:.1f} (kernel {tm.probe_kernel_ms:.1f}) reduce {tm.reduce_ms:.1f}; overlaps {st.verified_overlaps}", flush=True)
        ctx.close()
