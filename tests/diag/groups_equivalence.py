"""diagnostic: minimiser groups by the library's rule / forced off / forced on give the same edge list and counters (6 M reads at 10x and 25x coverage); index and probe times beside"""
import os, sys, zlib
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, fixtures as fx, sage2_amd as s2
n = 6_000_000
for cov in (10, 25):
    p = fx.synth_params(dict(seed=7, genome_len=int(n * 150 / cov), n_reads=n, read_len=150))
    out = []
    for mode in (None, "0", "1"):
        if mode is None: os.environ.pop("SAGE2OV_MINIMIZER_INDEX", None)
        else: os.environ["SAGE2OV_MINIMIZER_INDEX"] = mode
        c = s2.Context(40, device=0); c.reads_add_synth(p, s2.synth_genome(p)); c.reads_organize(); c.run_steps23()
        e = c.edges(); st = c.overlap_stats(); tm = c.timings()
        out.append((zlib.crc32(e.tobytes()), len(e), st.verified_overlaps, st.contained_extension)); print(cov, mode, out[-1], "index %.2f probe %.2f" % (tm.index_ms, tm.probe_ms), flush=True)
        c.close()
    assert out[0] == out[1] == out[2], "groups on / off / by rule differ"
print("EQUAL")
