"""diagnostic: configs[1] with 0.1 % substitution errors (bench.py's noisy_variant), a few whole steps with the library's phase times; SAGE2OV_TIMING=1 prints the reduce phase's laps"""
import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import fixtures as fx, sage2_amd as s2
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000; reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
p = fx.synth_params(dict(seed=2, genome_len=3 * n, n_reads=n, read_len=150, err_ppm=1000))
ctx = s2.Context(40, device=0); ctx.reads_add_synth(p, s2.synth_genome(p)); ctx.reads_organize()
for rep in range(reps):
    ctx.timings_reset(); ctx.run_steps23(); tm = ctx.timings(); st = ctx.overlap_stats()
    print(f"index {tm.index_ms:.2f} probe {tm.probe_ms:.2f} reciprocal {tm.reciprocal_ms:.2f} reduce {tm.reduce_ms:.2f} (marks {tm.reduce_marks_ms:.2f}) convert {tm.convert_ms:.2f} | edges {len(ctx.edges())} removed {st.transitive_removed} inserted {st.edges_inserted}")
