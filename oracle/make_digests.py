#!/usr/bin/env python3
"""oracle/make_digests.py -- TEST INFRASTRUCTURE: full-size digests from the pinned CPU restatement.

Runs in the build container only (minutes of CPU, tens of GB of RAM for c3).  For each named configuration of
tests/digests.py::CONFIGS it generates the synthetic reads with the repo's own generator, runs oracle/liboracle.so
(the restatement that tests/test_oracle_golden.py pins byte-for-byte to the reference binary's files) through steps 1-3 and
writes tests/golden/<name>_digest.json: counters and checksums of the per-read results, the canonical edge list and P.graph3.
The `-m gpu` full-size tests and bench.py compare the HIP path with these numbers.

Usage: python oracle/make_digests.py c1 c2_1m c2 c2_noisy c3      (default: all that are missing)
"""
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np          # noqa: E402
import digests as dg        # noqa: E402
import fixtures as fx       # noqa: E402
import oracle_lib as ol     # noqa: E402
import sage2_amd as s2      # noqa: E402  (generator only: sage2ov_synth_* are host functions of the library)


def make(name, threads):
    cfg = dg.CONFIGS[name]
    k, pd = cfg["k"], cfg["synth"]
    p = fx.synth_params(pd)
    t0 = time.time()
    g = s2.synth_genome(p)
    o = ol.Oracle(k, threads)
    step = 2_000_000
    for first in range(0, pd["n_reads"], step):                       # batches: the ASCII of 50 M reads is 7.5 GB
        n = min(step, pd["n_reads"] - first)
        bases, off = s2.synth_reads_ascii(p, g, first, n)
        o.add_reads_ascii(bases, off)
    del g
    print(f"[{name}] reads generated and staged: {time.time() - t0:.0f} s", flush=True)
    o.organize(); print(f"[{name}] organised: N = {o.counter('N')}  ({time.time() - t0:.0f} s)", flush=True)
    o.build_index(); print(f"[{name}] index built ({time.time() - t0:.0f} s)", flush=True)
    o.initial(); print(f"[{name}] initial pass done ({time.time() - t0:.0f} s)", flush=True)
    o.reduce(); print(f"[{name}] reduce done ({time.time() - t0:.0f} s)", flush=True)
    o.convert(); print(f"[{name}] convert done ({time.time() - t0:.0f} s)", flush=True)
    d = dict(name=name, k=k, synth=pd, generated_by="oracle/make_digests.py (oracle/liboracle.so)")
    r, l, s, c = o.export_initial()
    d.update(dg.initial_digest(r, l, s, c)); del r, l, s, c
    e = o.export_edges()
    d.update(dg.edges_digest(e[:, 0], e[:, 1], e[:, 2], e[:, 3], e[:, 4])); del e
    pk, ln, fr = o.export_reads()
    d.update(dg.reads_digest(pk, ln, fr, pd["read_len"])); del pk, ln, fr
    cn = o.counters()
    assert cn["n_ov"] == d["n_ov"] and cn["edges"] == d["edges"]
    d.update(n_unique=cn["N"], good_reads=cn["good_reads"], contained_extension=cn["contained"], contained_size=cn["contained_size"],
             left_to_explore=cn["N"] - cn["contained"] - cn["contained_size"], edges_inserted=cn["edges_inserted"],
             transitive_removed=cn["transitive_removed"], long_buckets=cn["long_buckets"], keys=cn["keys"])
    tmp = tempfile.mkdtemp(prefix="digest_", dir=os.environ.get("DIGEST_TMP", "/tmp"))
    gp = os.path.join(tmp, "t.graph3")
    o.write_graph3(gp); d.update(dg.file_digest(gp)); os.remove(gp); os.rmdir(tmp)
    ref = os.path.join(ROOT, "oracle", "_ref", "SAGE2")
    if pd["n_reads"] <= 1_000_000 and os.path.exists(ref):
        # small configurations: the reference binary itself on the same reads -- its P.graph3 must be the restatement's, byte for byte
        import subprocess, shutil
        t2 = tempfile.mkdtemp(prefix="digest_ref_", dir=os.environ.get("DIGEST_TMP", "/tmp"))
        fa = os.path.join(t2, "x.fa"); s2.synth_write_fasta(p, fa)
        subprocess.run([ref, "-f", fa, "-k", str(k), "-o", os.path.join(t2, "out"), "-p", "t", "-M", "3"], check=True, stdout=subprocess.DEVNULL,
                       env=dict(os.environ, OMP_NUM_THREADS=str(threads)))
        rd = dg.file_digest(os.path.join(t2, "out", "t.graph3")); shutil.rmtree(t2)
        assert rd["graph3_md5"] == d["graph3_md5"], "the reference binary's P.graph3 differs from the restatement's"
        d["reference_binary_graph3_identical"] = True
    d["oracle_seconds"] = dict(index=o.time(0), initial=o.time(1), reduce=o.time(2), convert=o.time(3), total_wall=time.time() - t0, threads=threads)
    o.close()
    json.dump(d, open(dg.path_of(name), "w"), indent=1)
    print(f"[{name}] written: {json.dumps({kk: d[kk] for kk in ('n_unique', 'n_ov', 'edges', 'edges_crc32', 'graph3_md5')})}", flush=True)


if __name__ == "__main__":
    names = sys.argv[1:] or [n for n in dg.CONFIGS if not os.path.exists(dg.path_of(n))]
    threads = int(os.environ.get("DIGEST_THREADS", "8"))
    for nm in names:
        make(nm, threads)
