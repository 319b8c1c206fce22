"""diagnostic: do the minimiser groups pay at LOW coverage (short runs of shifted reads: more reads without a predecessor)?  index build + probe pass with and without them"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import fixtures as fx, sage2_amd as s2
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40_000_000
for cov in [float(x) for x in (sys.argv[2:] or ["10", "20", "50"])]:
    p = fx.synth_params(dict(seed=3, genome_len=int(n * 150 / cov), n_reads=n, read_len=150))
    ctx = s2.Context(40, device=0); ctx.reads_add_synth(p, s2.synth_genome(p)); ctx.reads_organize()
    for mi in ("0", "1"):
        os.environ["SAGE2OV_MINIMIZER_INDEX"] = mi
        for rep in range(2):
            ctx.timings_reset(); ctx.index_build(); ctx.overlap_probe_shard(); tm = ctx.timings()
        print(f"coverage {cov:.0f}x reads {n} unique {ctx.reads_stats().unique_reads} groups {mi}: index {tm.index_ms:.2f} probe {tm.probe_ms:.2f} sum {tm.index_ms + tm.probe_ms:.2f}", flush=True)
    ctx.close()
