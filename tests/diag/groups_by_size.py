"""diagnostic: do the minimiser groups pay at this size?  index build + probe pass with and without them (SAGE2OV_TIMING=1 prints the group table's load)"""
import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import fixtures as fx, sage2_amd as s2
for n in [int(x) for x in sys.argv[1:]]:
    p = fx.synth_params(dict(seed=3, genome_len=3 * n, n_reads=n, read_len=150))
    ctx = s2.Context(40, device=0); ctx.reads_add_synth(p, s2.synth_genome(p)); ctx.reads_organize()
    for mi in ("0", "1"):
        os.environ["SAGE2OV_MINIMIZER_INDEX"] = mi
        for rep in range(2):
            ctx.timings_reset(); ctx.index_build(); ctx.overlap_probe_shard(); tm = ctx.timings()
        print(f"reads {n} unique {ctx.reads_stats().unique_reads} groups {mi}: index {tm.index_ms:.2f} probe {tm.probe_ms:.2f} sum {tm.index_ms + tm.probe_ms:.2f}", flush=True)
    ctx.close()
