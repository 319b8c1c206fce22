"""diagnostic: 150-bp reads with k = 20..23 (129-131 windows: the three-windows-per-lane instantiation): timing, and parity against the oracle"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import fixtures as fx, sage2_amd as s2, oracle_lib as ol, numpy as np
n = 2_000_000
pd = dict(seed=5, genome_len=n * 150 // 50, n_reads=n, read_len=150)
bases, off = fx.make_reads(pd)
for k in (20, 21, 22, 23):
    ctx = s2.Context(k); ctx.reads_add_ascii(bases, off); ctx.reads_organize(); ctx.run_steps23()
    t0 = time.time(); ctx.run_steps23(); t1 = time.time(); tm = ctx.timings()
    print(f"k {k}: {1e3*(t1-t0):.1f} ms kernel {tm.probe_kernel_ms:.1f} sequential {tm.sequential_reads} edges {ctx.overlap_stats().edges}", flush=True)
    ctx.close()
# parity against the oracle at k=21 (130 windows: the three-windows-per-lane instantiation), clean and noisy
for err in (0, 2000):
    pd = dict(seed=9, genome_len=300000, n_reads=100000, read_len=150, err_ppm=err)
    b, o = fx.make_reads(pd)
    c = s2.Context(21); c.reads_add_ascii(b, o); c.reads_organize(); c.run_steps23()
    orc = ol.Oracle(21, 8); orc.add_reads_ascii(b, o); orc.organize(); orc.run_all()
    e, oe = c.edges(), orc.export_edges()
    ok = len(e) == len(oe) and np.array_equal(e["from"], oe[:, 0]) and np.array_equal(e["to"], oe[:, 1]) and np.array_equal(e["type"], oe[:, 2]) and np.array_equal(e["length"], oe[:, 3])
    r, l, st, cn = c.overlap_export_initial(); orr, orl, ost, ocn = orc.export_initial()
    # (index 0 is unused; the oracle's status array has been advanced by its BFS, so only records and connections are compared here)
    print("err", err, "edges equal", ok, "extension records and connections equal", np.array_equal(r[1:], orr[1:]) and np.array_equal(l[1:], orl[1:]) and np.array_equal(cn, ocn), flush=True)
    c.close(); orc.close()
