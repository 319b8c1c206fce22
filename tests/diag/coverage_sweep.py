"""diagnostic: step time per read at several coverages (150-bp reads, k = 40, error-free): candidates per read grow with the coverage; beyond 128 the fast kernel hands a read over"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import fixtures as fx, sage2_amd as s2
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
for cov in (20, 35, 50, 70, 90, 120):
    p = fx.synth_params(dict(seed=7, genome_len=n * 150 // cov, n_reads=n, read_len=150))
    ctx = s2.Context(40, device=0); ctx.reads_add_synth(p, s2.synth_genome(p)); ctx.reads_organize(); ctx.run_steps23()
    ctx.timings_reset(); t0 = time.time(); ctx.run_steps23(); t1 = time.time(); tm = ctx.timings(); st = ctx.overlap_stats(); u = ctx.reads_stats().unique_reads
    print(f"coverage {cov}: unique {u}, {1e3 * (t1 - t0):.1f} ms ({1e9 * (t1 - t0) / u:.2f} ns/read) index {tm.index_ms:.1f} probe {tm.probe_ms:.1f} (kernel {tm.probe_kernel_ms:.1f}, handed over {tm.sequential_reads}) "
          f"reduce {tm.reduce_ms:.1f}; overlaps {st.verified_overlaps} ({st.verified_overlaps / u:.1f} per read), {st.verified_overlaps / (t1 - t0) / 1e9:.2f} G overlaps/s", flush=True)
    ctx.close()
