"""diagnostic experiment at full size: probe kernel time with the read store in id order (the reference's: lexicographic, random with respect to the
genome) against a store permuted into locality order on the device (SAGE2OV_EXPERIMENT_LOCALITY_IDS: ids become positions -- results are not
the reference's, only the timing means something).  usage: locality_ids_device.py <reads> [order bits]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
if len(sys.argv) > 2: os.environ["SAGE2OV_ORDER_BITS"] = sys.argv[2]
import fixtures as fx, sage2_amd as s2
p = fx.synth_params(dict(seed=3 if n == 50_000_000 else 2, genome_len=3 * n, n_reads=n, read_len=150))
g = s2.synth_genome(p)
for tag, env in (("id order", None), ("locality ids", "1"), ("locality ids, plain processing order", "2")):
    if env: os.environ["SAGE2OV_EXPERIMENT_LOCALITY_IDS"] = "1"
    if env == "2": os.environ["SAGE2OV_NO_LOCALITY"] = "1"
    ctx = s2.Context(40, device=0); ctx.reads_add_synth(p, g); ctx.reads_organize()
    ctx.run_steps23(); ctx.run_steps23(); t = ctx.timings(); N = ctx.reads_stats().unique_reads
    print(f"{tag:40s} kernel {t.probe_kernel_ms / t.probe_kernel_launches:8.2f} ms  {1e6 * t.probe_kernel_ms / t.probe_kernel_launches / N:.3f} ns/read  probe phase {t.probe_ms:.1f} index {t.index_ms:.1f} step {t.total_ms:.1f} ms  overlaps {ctx.overlap_stats().verified_overlaps}", flush=True)
    ctx.close()
