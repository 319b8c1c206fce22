"""diagnostic: phase timing of the 0.1 %-error variant of configs[1] (SAGE2OV_TIMING=1 prints the parts of the reduce phase)"""
import sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import fixtures as fx, sage2_amd as s2
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
err = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
p = fx.synth_params(dict(seed=2, genome_len=3 * n, n_reads=n, read_len=150, err_ppm=err))
ctx = s2.Context(40, device=0); ctx.reads_add_synth(p, s2.synth_genome(p)); ctx.reads_organize()
ctx.run_steps23()
for rep in range(2):
    ctx.timings_reset()
    t0 = time.perf_counter(); ctx.index_build(); t1 = time.perf_counter(); ctx.overlap_initial(); t2 = time.perf_counter(); ctx.overlap_reduce(); t3 = time.perf_counter(); ctx.overlap_convert(); t4 = time.perf_counter()
    tm = ctx.timings()
    print(f"wall: index {1e3*(t1-t0):.2f} initial {1e3*(t2-t1):.2f} reduce {1e3*(t3-t2):.2f} convert {1e3*(t4-t3):.2f} total {1e3*(t4-t0):.2f} | events: index {tm.index_ms:.2f} probe {tm.probe_ms:.2f} (kernel {tm.probe_kernel_ms:.2f}) recip {tm.reciprocal_ms:.2f} reduce {tm.reduce_ms:.2f} convert {tm.convert_ms:.2f}", flush=True)
