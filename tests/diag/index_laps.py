import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import fixtures as fx, sage2_amd as s2
n = 50_000_000
p = fx.synth_params(dict(seed=3, genome_len=3 * n, n_reads=n, read_len=150))
ctx = s2.Context(40, device=0); ctx.reads_add_synth(p, s2.synth_genome(p)); ctx.reads_organize()
for rep in range(3):
    ctx.timings_reset(); ctx.index_build(); print("index_ms", ctx.timings().index_ms, flush=True)
