"""diagnostic: random data-set shapes through steps 1-3 on the device against the CPU oracle: per-read extension records, connections,
edge list, counters"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, fixtures as fx, sage2_amd as s2, oracle_lib as ol
n0, n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 30
scale = int(sys.argv[3]) if len(sys.argv) > 3 else 1                  # multiplies genome length and read count
mid_reads = len(sys.argv) > 4 and sys.argv[4] == "mid"                # 161..251 bases: the 16-dword instantiations of the fast kernel, with errors and repeats
long_reads = len(sys.argv) > 4 and sys.argv[4] == "long"              # read lengths of the 16- and 32-word layouts (sequential kernel), fewer reads
bad = 0
for seed in range(n0, n1):
    r = np.random.default_rng(seed)
    L = int(r.choice([60, 100, 123, 124, 150, 160, 161, 200, 250, 251, 252, 300])); k = int(r.choice([16, 21, 31, 40, 55, 64, 70]))
    if mid_reads: L = int(r.choice([161, 170, 185, 200, 225, 250, 251])); k = int(r.choice([21, 31, 40, 55, 70, 96]))
    if long_reads: L = int(r.choice([300, 400, 504, 505, 600, 750, 900, 992, 993, 1018])); k = int(r.choice([21, 40, 55, 70, 96, 127]))
    if k >= L - 5: k = max(16, L // 2)
    pd = dict(seed=5000 + seed, genome_len=scale * int(r.integers(20000, 120000)), n_reads=scale * (int(r.integers(2500, 7000)) if long_reads else int(r.integers(14000, 40000))), read_len=L, err_ppm=int(r.choice([300, 1500, 5000]) if mid_reads else r.choice([0, 300, 1500, 5000])),
              n_repeat_families=int(r.integers(0, 5)), repeat_copies=int(r.integers(20, 300)), repeat_len=int(r.integers(100, 500)))
    if r.random() < 0.35: pd["read_len_min"] = max(k + 2, L - int(r.integers(5, 60)))
    if long_reads and r.random() < 0.5: pd["read_len_min"] = max(k + 2, L - int(r.integers(50, 450)))
    bases, off = fx.make_reads(pd)
    g = s2.Context(k); g.reads_add_ascii(bases, off); g.reads_organize(); g.run_steps23()
    o = ol.Oracle(k, 16); o.add_reads_ascii(bases, off); o.organize(); o.run_all()
    gr, gl, gs, gc = g.overlap_export_initial(); orr, orl, ors, orc = o.export_initial()
    e, oe = g.edges(), o.export_edges(); st = g.overlap_stats()
    ok = np.array_equal(gc, orc) and np.array_equal(gr[1:], orr[1:]) and np.array_equal(gl[1:], orl[1:]) and len(e) == len(oe) and np.array_equal(e["from"], oe[:, 0]) \
        and np.array_equal(e["to"], oe[:, 1]) and np.array_equal(e["type"], oe[:, 2]) and np.array_equal(e["length"], oe[:, 3]) and np.array_equal(e["length_twin"], oe[:, 4]) \
        and (st.edges_inserted, st.transitive_removed, st.verified_overlaps) == (o.counter("edges_inserted"), o.counter("transitive_removed"), o.counter("n_ov"))
    if not ok:
        bad += 1; print("MISMATCH seed", seed, pd, "k", k, flush=True)
        print("   conn", np.array_equal(gc, orc), "right", np.array_equal(gr[1:], orr[1:]), "left", np.array_equal(gl[1:], orl[1:]), "edges", len(e), len(oe),
              "counters", (st.edges_inserted, st.transitive_removed, st.verified_overlaps), (o.counter("edges_inserted"), o.counter("transitive_removed"), o.counter("n_ov")),
              "records differing", int((gr[1:] != orr[1:]).sum()), int((gl[1:] != orl[1:]).sum()), int((gc != orc).sum()),
              "initial classes differing", int((np.where(np.isin(gs[1:], (1, 2)), 0, gs[1:]) != np.where(np.isin(ors[1:], (1, 2)), 0, ors[1:])).sum()),
              "unresolved", st.left_to_explore, int((ors[1:] == 0).sum() + np.isin(ors[1:], (1, 2)).sum()), flush=True)
    g.close(); o.close()
print("data sets", n1 - n0, "mismatches", bad, flush=True)
