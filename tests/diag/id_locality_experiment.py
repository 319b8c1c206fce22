"""diagnostic experiment: how fast would the probe kernel be if a read's neighbours were adjacent in the read store?
The unique reads are renumbered by (global minimiser, position of the minimiser) -- NOT the reference's ids, results are not comparable --
and imported as a ready-made read store; the kernel time per read is compared with the normal id order."""
import os, sys, time
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import fixtures as fx, sage2_amd as s2
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
p = fx.synth_params(dict(seed=2, genome_len=3 * n, n_reads=n, read_len=150))
ctx = s2.Context(40, device=0); ctx.reads_add_synth(p, s2.synth_genome(p)); ctx.reads_organize()
st = ctx.reads_stats(); N, S = st.unique_reads, st.words_per_read
def run(c, tag):
    c.run_steps23(); c.run_steps23(); t = c.timings()
    print(tag, "kernel %.2f ms  %.3f ns/read  step %.2f ms  edges %d" % (t.probe_kernel_ms / t.probe_kernel_launches, 1e6 * t.probe_kernel_ms / t.probe_kernel_launches / N, t.total_ms, c.overlap_stats().edges))
run(ctx, "ids by content (reference order):")
words, freq = ctx.reads_export_words(); ctx.close()
W = words.reshape(N + 1, S)
# bases of every read as 2-bit codes (150 columns), from the big-endian words
L = 150
codes = np.zeros((N + 1, L), dtype=np.uint8)
for c in range((L + 31) // 32):
    w = W[:, c]
    for b in range(32):
        pos = 32 * c + b
        if pos < L: codes[:, pos] = (w >> np.uint64(62 - 2 * b)) & np.uint64(3)
# canonical 16-mers, hashed; global minimiser and its offset
k = 16
f = np.zeros((N + 1, L - k + 1), dtype=np.uint64); r = np.zeros_like(f)
for j in range(k):
    f = (f << np.uint64(2)) | codes[:, j:j + L - k + 1].astype(np.uint64)
    r = r | ((np.uint64(3) - codes[:, j:j + L - k + 1].astype(np.uint64)) << np.uint64(2 * j))
can = np.minimum(f, r)
h = (can * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(20)
arg = np.argmin(h, axis=1); mh = h[np.arange(N + 1), arg]
fwd = f[np.arange(N + 1), arg] <= r[np.arange(N + 1), arg]
rel = np.where(fwd, arg, (L - k) - arg)
key = (mh << np.uint64(9)) | rel.astype(np.uint64)
order = np.argsort(key[1:], kind="stable") + 1
W2 = np.zeros_like(W); W2[1:] = W[order]
ctx2 = s2.Context(40, device=0)
ctx2.reads_import_words(W2.reshape(-1), N, S, st.max_read_length, np.ones(N + 1, dtype=np.uint16), st.good_reads, st.total_bp)
run(ctx2, "ids by locality (experiment):    ")
os.environ["SAGE2OV_NO_LOCALITY"] = "1"
ctx3 = s2.Context(40, device=0)
ctx3.reads_import_words(W2.reshape(-1), N, S, st.max_read_length, np.ones(N + 1, dtype=np.uint16), st.good_reads, st.total_bp)
run(ctx3, "ids by locality, plain id order:  ")
