"""diagnostic: step time per read over k (minimum overlap) for 100 and 150 bp reads, clean, to spot slow paths"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import fixtures as fx, sage2_amd as s2
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
for L in (100, 150):
    pd = dict(seed=5, genome_len=n * L // 50, n_reads=n, read_len=L)
    bases, off = fx.make_reads(pd)
    for k in (16, 21, 31, 40, 55, 63, 70, 90):
        if k >= L: continue
        ctx = s2.Context(k); ctx.reads_add_ascii(bases, off); ctx.reads_organize(); ctx.run_steps23()
        t0 = time.time(); ctx.run_steps23(); t1 = time.time(); tm = ctx.timings(); st = ctx.overlap_stats()
        print(f"L {L} k {k}: {1e3 * (t1 - t0):.1f} ms ({1e9 * (t1 - t0) / ctx.reads_stats().unique_reads:.1f} ns/read) index {tm.index_ms:.1f} probe {tm.probe_ms:.1f} (kernel {tm.probe_kernel_ms:.1f}, sequential reads {tm.sequential_reads}) "
              f"reduce {tm.reduce_ms:.1f}; overlaps {st.verified_overlaps}, unresolved {st.left_to_explore}, long buckets {ctx.index_stats().long_buckets}", flush=True)
        ctx.close()
