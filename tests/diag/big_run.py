"""diagnostic (capacity, BASELINE configs[4]): the largest error-free read set one MI355X holds, through steps 1-3, checked by size-independent
properties (one path through all unique reads: N-1 edges, at most one neighbour per read end, two free ends, strictly sorted canonical list) and
timed; prints the device-memory high-water mark the library sampled (sage2ov_debug_meminfo).
usage: python tests/diag/big_run.py <n_reads> [k=55] [steps=2] [read_len=150]     (genome = 3 x n_reads: 50x coverage; seed 5, SURVEY 8d's C5)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import fixtures as fx, sage2_amd as s2

n_reads = int(sys.argv[1]); k = int(sys.argv[2]) if len(sys.argv) > 2 else 55; steps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
L = int(sys.argv[4]) if len(sys.argv) > 4 else 150
pd = dict(seed=5, genome_len=3 * n_reads * L // 150, n_reads=n_reads, read_len=L)
p = fx.synth_params(pd)
t0 = time.time()
ctx = s2.Context(k, device=0)
g = s2.synth_genome(p)
step = 50_000_000
for first in range(0, n_reads, step):                        # staged in slices: progress lines for the watchdog
    ctx.reads_add_synth(p, g, first, min(step, n_reads - first)); print(f"[big] staged {min(first + step, n_reads)} reads ({time.time() - t0:.0f} s)", flush=True)
del g
ctx.reads_organize(); st = ctx.reads_stats()
print(f"[big] organised: {st.good_reads} good -> {st.unique_reads} unique ({time.time() - t0:.0f} s; device {ctx.timings().organize_ms:.0f} ms)", flush=True)
out = dict(workload=f"{n_reads} x {L} bp, k={k}, genome {pd['genome_len']}, seed 5, error-free", unique_reads=st.unique_reads, steps=[])
for s_ in range(steps):
    t1 = time.perf_counter(); ctx.run_steps23(); dt = time.perf_counter() - t1; tm = ctx.timings(); o = ctx.overlap_stats(); mi = ctx.debug_meminfo()
    out["steps"].append(dict(ms=1e3 * dt, index_ms=tm.index_ms, probe_ms=tm.probe_ms, reciprocal_ms=tm.reciprocal_ms, reduce_ms=tm.reduce_ms, convert_ms=tm.convert_ms,
                             overlaps=o.verified_overlaps, edges=o.edges, overlaps_per_s=o.verified_overlaps / dt,
                             mem_high_water_GB=(mi["total"] - mi["lowest_free"]) / 1e9, mem_total_GB=mi["total"] / 1e9, arena_GB=mi["arena"] / 1e9))
    print("[big]", json.dumps(out["steps"][-1]), flush=True)
print(f"[big] downloading {ctx.overlap_stats().edges} edges for the property checks ({time.time() - t0:.0f} s)", flush=True)
n = st.unique_reads
e = ctx.edges()
f, t, ty = e["from"].astype(np.int64), e["to"].astype(np.int64), e["type"].astype(np.int64)
ok = dict(edges_n_minus_1=bool(len(e) == n - 1), from_lt_to=bool(np.all(f < t) and np.all(f >= 1) and np.all(t <= n)))
key = (f.astype(np.uint64) << np.uint64(34)) | (t.astype(np.uint64) << np.uint64(2)) | ty.astype(np.uint64)      # (ids reach 2^30: unsigned)
ok["strictly_sorted"] = bool(np.all(key[1:] > key[:-1])); del key
ok["lengths_in_range"] = bool(np.all(e["length"] > 0) and np.all(e["length"] < L) and np.all(e["length_twin"] > 0) and np.all(e["length_twin"] < L))
src_end = (ty >> 1) & 1; dst_end = 1 - (ty & 1)
use = np.bincount(2 * f + src_end, minlength=2 * (n + 1)) + np.bincount(2 * t + dst_end, minlength=2 * (n + 1))
ok["one_neighbour_per_end"] = bool(use[2:].max() == 1); ok["two_free_ends"] = bool(int((use[2:] == 0).sum()) == 2)
out["properties"] = ok; out["all_ok"] = all(ok.values())
print(json.dumps(out), flush=True)
ctx.close()
sys.exit(0 if out["all_ok"] else 1)
