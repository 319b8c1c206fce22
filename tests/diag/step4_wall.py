"""diagnostic: wall time of step 4 by part (SAGE2OV_TIMING=1) on the graph of a clean read set"""
import sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import fixtures as fx, sage2_amd as s2
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
p = fx.synth_params(dict(seed=3 if n == 50_000_000 else 2, genome_len=3 * n, n_reads=n, read_len=150))
ctx = s2.Context(40, device=0); ctx.reads_add_synth(p, s2.synth_genome(p)); ctx.reads_organize(); ctx.run_steps23()
for rep in range(2):
    t = time.perf_counter(); ctx.graph_simplify(); print(f"graph_simplify wall {1e3 * (time.perf_counter() - t):.1f} ms, device {ctx.simplify_stats().device_ms:.1f} ms", flush=True)
