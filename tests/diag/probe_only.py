"""diagnostic: index build once, then the probe kernel alone a few times (results are not consumed: usable with the cut / stamp builds)"""
import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import fixtures as fx, sage2_amd as s2
n = int(sys.argv[1]); reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
p = fx.synth_params(dict(seed=3 if n == 50_000_000 else 2, genome_len=3 * n, n_reads=n, read_len=150))
ctx = s2.Context(40, device=0); ctx.reads_add_synth(p, s2.synth_genome(p)); ctx.reads_organize()
ctx.index_build()
for rep in range(reps):
    ctx.timings_reset(); ctx.overlap_probe_shard(); tm = ctx.timings()
    print(f"probe kernel {tm.probe_kernel_ms:.3f} ms, unique reads {ctx.reads_stats().unique_reads}")
