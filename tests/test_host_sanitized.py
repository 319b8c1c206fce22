"""CPU: the host side of the product -- the mapped, chunk-parallel FASTA / FASTQ readers, step 1 on a device-less context, the pipelined text writer and the
P.reads loader -- rebuilt with -fsanitize=address,undefined (sage2_amd/csrc/Makefile: libsage2ov_hostasan.so) and driven in a child interpreter with libasan
preloaded: an out-of-bounds access (the parsers walk a mapping with memchr and eight-byte loads), a use-after-free or undefined behaviour aborts the child.
The P.reads file it writes must still be the reference's."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import fixtures as fx, sage2_amd as s2
tmp = sys.argv[2]
m = fx.golden("g6_k70_150")
fa = os.path.join(tmp, "x.fa"); s2.synth_write_fasta(fx.synth_params(m["synth"]), fa)
c = s2.Context(m["k"], device=-2); c.reads_add_file(fa); c.reads_organize()
rp = os.path.join(tmp, "t.reads"); c.reads_save(rp)
assert fx.md5_file(rp) == m["reads_md5"]
a = c.reads_export(); c.close()
d = s2.Context(m["k"], device=-2); d.reads_load(rp); b = d.reads_export(); d.close()          # the chunk-parallel loader
assert all(np.array_equal(x, y) for x, y in zip(a, b))
# a big single-line FASTA with every awkward shape near chunk borders, a four-line FASTQ, and two mate files: parallel readers against the sequential one
rng = np.random.default_rng(1)
def seq(): return "".join(rng.choice(list("ACGTacgtN"), p=[.2475] * 4 + [.00225] * 4 + [.001], size=int(rng.integers(75, 131))))
big = os.path.join(tmp, "big.fa")
with open(big, "w", newline="") as f:
    for i in range(16000):
        s = seq()
        f.write(f">r{i}\n{s}\n" if i % 911 else (f">r{i} two lines\r\n{s[:33]}\r\n{s[33:]}\r\n" if i % 2 else f">r{i}\n{s[:20]} {s[20:]}\n"))
fq = os.path.join(tmp, "big.fq")
with open(fq, "w") as f:
    for i in range(14000):
        s = seq(); f.write(f"@q{i}\n{s}\n+\n{'@' + 'I' * (len(s) - 1)}\n")
m1, m2 = os.path.join(tmp, "m_1.fa"), os.path.join(tmp, "m_2.fa")
for path, cnt in ((m1, 15001), (m2, 15000)):
    with open(path, "w") as f:
        for i in range(cnt):
            f.write(f">p{i}\n{seq()}\n")
for files in ((big, None), (fq, None), (m1, m2)):
    outs = []
    for sequential in (False, True):
        if sequential: os.environ["SAGE2OV_SEQUENTIAL_READER"] = "1"
        else: os.environ.pop("SAGE2OV_SEQUENTIAL_READER", None)
        c = s2.Context(21, device=-2); c.reads_add_file(*files) if files[1] else c.reads_add_file(files[0]); c.reads_organize()
        st = c.reads_stats(); outs.append((c.reads_export(), (st.total_reads, st.good_reads, st.unique_reads, st.total_bp))); c.close()
    assert outs[0][1] == outs[1][1] and all(np.array_equal(x, y) for x, y in zip(outs[0][0], outs[1][0])), files
print("SAN_OK")
'''


def test_host_side_under_asan_ubsan(tmp_path):
    so = os.path.join(ROOT, "sage2_amd", "libsage2ov_hostasan.so")
    if not os.path.exists(os.path.join(ROOT, "sage2_amd", "csrc", "sage2ov_device.o")):
        pytest.skip("product not built here (the sanitized host objects link with its device object)")
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "sage2_amd", "csrc"), "../libsage2ov_hostasan.so"], capture_output=True, text=True)
    if r.returncode != 0 or not os.path.exists(so):
        pytest.skip("sanitizer build not available: " + r.stderr[-300:])
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    ubsan = subprocess.run(["gcc", "-print-file-name=libubsan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan) or not os.path.isabs(ubsan):
        pytest.skip("libasan / libubsan not found")
    env = dict(os.environ, LD_PRELOAD=asan + " " + ubsan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               OMP_NUM_THREADS="4", SAGE2OV_LIB=so)
    env.pop("SAGE2OV_SEQUENTIAL_READER", None)
    r = subprocess.run([sys.executable, "-c", CHILD, ROOT, str(tmp_path)], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0 and "SAN_OK" in r.stdout, (r.stdout[-2000:] + r.stderr[-4000:])
