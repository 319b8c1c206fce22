"""diagnostic: wall-clock of the CLI (sage2_amd/sage2ov -M 3, or -M <argv[2]>) on a synthetic FASTA, by phase from its log"""
import os, sys, subprocess, tempfile, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import fixtures as fx, sage2_amd as s2
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
max_step = sys.argv[2] if len(sys.argv) > 2 else "3"
p = fx.synth_params(dict(seed=2, genome_len=3 * n, n_reads=n, read_len=150))
tmp = tempfile.mkdtemp(); fa = os.path.join(tmp, "x.fa"); out = os.path.join(tmp, "out")
t0 = time.time(); s2.synth_write_fasta(p, fa); t1 = time.time()
print("fasta written: %.1f s, %.0f MB" % (t1 - t0, os.path.getsize(fa) / 1e6))
t0 = time.time()
subprocess.run([os.path.join(R, "sage2_amd", "sage2ov"), "-f", fa, "-k", "40", "-o", out, "-p", "t", "-M", max_step], check=True, env=dict(os.environ, SAGE2OV_TIMING="1"))
print("CLI total: %.2f s" % (time.time() - t0))
log = open(os.path.join(out, "t.log")).read()
print("\n".join(l for l in log.splitlines() if "ime" in l or "sec" in l or "written" in l)[:1500])
