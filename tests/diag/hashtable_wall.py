"""diagnostic: wall time of `sage2ov -M 2 -s` (P.reads + P.hashTable, the host replay of the reference's serial insertion) on a synthetic FASTA"""
import os, sys, subprocess, tempfile, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import fixtures as fx, sage2_amd as s2
n = int(sys.argv[1])
p = fx.synth_params(dict(seed=2, genome_len=3 * n, n_reads=n, read_len=150))
tmp = tempfile.mkdtemp(); fa = os.path.join(tmp, "x.fa"); out = os.path.join(tmp, "out")
s2.synth_write_fasta(p, fa)
t0 = time.time(); subprocess.run([os.path.join(R, "sage2_amd", "sage2ov"), "-f", fa, "-k", "40", "-o", out, "-p", "t", "-M", "2", "-s"], check=True); print("CLI -M 2 -s: %.2f s" % (time.time() - t0))
print("".join(l for l in open(os.path.join(out, "t.log")) if "sec" in l)); print(os.path.getsize(os.path.join(out, "t.hashTable")) / 1e6, "MB")
