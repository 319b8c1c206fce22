"""diagnostic: random data-set shapes (read length, errors, repeat families, k): the device reduce (symmetric or ranked form, forced for any
number of unresolved reads) against the serial replay on the host -- edges and counters"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, fixtures as fx, sage2_amd as s2
n0, n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = 0; ranked = 0
for seed in range(n0, n1):
    r = np.random.default_rng(seed)
    L = int(r.choice([100, 150, 250])); k = int(r.choice([21, 31, 40, 55]))
    pd = dict(seed=1000 + seed, genome_len=int(r.integers(100000, 400000)), n_reads=int(r.integers(40000, 90000)), read_len=L, err_ppm=int(r.choice([0, 500, 2000, 5000])),
              n_repeat_families=int(r.integers(0, 9)), repeat_copies=int(r.integers(50, 600)), repeat_len=int(r.integers(100, 600)))
    if r.random() < 0.3: pd["read_len_min"] = L - int(r.integers(10, 50))
    bases, off = fx.make_reads(pd)
    res = {}
    for mode in ("device", "host"):
        os.environ.pop("SAGE2OV_HOST_REDUCE", None); os.environ.pop("SAGE2OV_DEVICE_REDUCE_MIN", None)
        if mode == "host": os.environ["SAGE2OV_HOST_REDUCE"] = "1"
        else: os.environ["SAGE2OV_DEVICE_REDUCE_MIN"] = "1"
        ctx = s2.Context(k); ctx.reads_add_ascii(bases, off); ctx.reads_organize(); ctx.run_steps23()
        st = ctx.overlap_stats(); res[mode] = (ctx.edges().tobytes(), st.edges_inserted, st.transitive_removed, st.left_to_explore); lb = ctx.index_stats().long_buckets
        ctx.close()
    ranked += lb > 0
    if res["device"] != res["host"]:
        bad += 1; print("MISMATCH seed", seed, pd, "k", k, res["device"][1:], res["host"][1:], flush=True)
print("data sets", n1 - n0, "with long buckets", ranked, "mismatches", bad, flush=True)
