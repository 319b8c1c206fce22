"""diagnostic: the probe pass with run mode (default) and without (SAGE2OV_NO_RUN_MODE=1) on the same context -- which reads' records differ, and how.
usage: python tests/diag/run_mode_diff.py <reads> [k] ; SAGE2OV_MINIMIZER_INDEX etc. are taken from the environment"""
import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import fixtures as fx, sage2_amd as s2
n = int(sys.argv[1]); k = int(sys.argv[2]) if len(sys.argv) > 2 else 40
p = fx.synth_params(dict(seed=3 if n == 50_000_000 else 2, genome_len=3 * n, n_reads=n, read_len=150))
def run(norun):
    if norun: os.environ["SAGE2OV_NO_RUN_MODE"] = "1"
    else: os.environ.pop("SAGE2OV_NO_RUN_MODE", None)
    ctx = s2.Context(k, device=0); ctx.reads_add_synth(p, s2.synth_genome(p)); ctx.reads_organize()
    ctx.index_build(); ctx.overlap_initial()
    r, l, s, c = ctx.overlap_export_initial(); ctx.close()
    return r.copy(), l.copy(), c.copy()
a = run(False); b = run(True)
for nm, x, y in zip(("right", "left", "conn"), a, b):
    d = np.nonzero(x != y)[0]
    print(nm, "differs for", len(d), "reads; first:", d[:20].tolist())
    for i in d[:12]:
        if nm == "conn": print("   id", int(i), "run", int(x[i]), "general", int(y[i]))
        else: print("   id", int(i), "run: id %d type %d len %d" % (x[i] & ((1 << 40) - 1), (x[i] >> 40) & 3, x[i] >> 42), "| general: id %d type %d len %d" % (y[i] & ((1 << 40) - 1), (y[i] >> 40) & 3, y[i] >> 42))
