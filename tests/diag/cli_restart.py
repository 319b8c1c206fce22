"""diagnostic: wall-clock of the CLI's restart paths on a synthetic FASTA -- `-m 2 -M 3` (loads P.reads) and `-m 4 -M 4` (loads P.reads and P.graph3)"""
import os, sys, subprocess, tempfile, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import fixtures as fx, sage2_amd as s2
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
p = fx.synth_params(dict(seed=2, genome_len=3 * n, n_reads=n, read_len=150))
tmp = tempfile.mkdtemp(); fa = os.path.join(tmp, "x.fa"); out = os.path.join(tmp, "out")
s2.synth_write_fasta(p, fa)
cli = os.path.join(R, "sage2_amd", "sage2ov"); env = dict(os.environ, SAGE2OV_TIMING="1")
def run(*a):
    t0 = time.time(); subprocess.run([cli, "-f", fa, "-k", "40", "-o", out, *a], check=True, env=env, stderr=subprocess.DEVNULL); return time.time() - t0
print("-M 3            : %.2f s" % run("-p", "t", "-M", "3"))
print("-m 2 -M 3 (-i t): %.2f s" % run("-p", "u", "-i", "t", "-m", "2", "-M", "3"))
print("-m 4 -M 4 (-i t): %.2f s" % run("-p", "v", "-i", "t", "-m", "4", "-M", "4"))
a, b = open(os.path.join(out, "t.graph3"), "rb").read().split(b"\n", 3), open(os.path.join(out, "u.graph3"), "rb").read().split(b"\n", 3)
print("edge records of the restarted run identical:", a[3] == b[3])
for pfx in ("u", "v"):
    print(pfx, "|", " | ".join(l.strip() for l in open(os.path.join(out, pfx + ".log")) if "sec" in l))
