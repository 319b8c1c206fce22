"""diagnostic: run with a -DSAGE2OV_STAMPS build (SAGE2OV_LIB=...): share of wave cycles per stage of the probe kernel, 100- and 150-bp reads"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import fixtures as fx, sage2_amd as s2
n = 2_000_000
for L in (100, 150):
    pd = dict(seed=5, genome_len=n * L // 50, n_reads=n, read_len=L)
    bases, off = fx.make_reads(pd)
    ctx = s2.Context(40); ctx.reads_add_ascii(bases, off); ctx.reads_organize(); ctx.run_steps23(); print("L", L, "unique", ctx.reads_stats().unique_reads, flush=True); ctx.run_steps23(); ctx.close()
