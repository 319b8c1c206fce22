#!/usr/bin/env python3
"""oracle/pin_reference.py -- TEST INFRASTRUCTURE: pin a full-size digest on the REFERENCE BINARY itself.

oracle/make_digests.py writes tests/golden/<name>_digest.json from the CPU restatement.  This script runs the reference binary
(oracle/_ref/SAGE2, compiled from /root/reference in place by oracle/Makefile) on the very same synthetic reads --
`SAGE2 -f x.fa -k K -o out -p t -M 3 -s` (main.cpp:37-132) -- and adds to the digest

    reference_binary: {graph3_md5, graph3_bytes, reads_md5, reads_bytes, counters (the log's "Total contained by extension" ...
                       economyGraph.cpp:485-487,569-571), seconds per function as the log prints them (1-s resolution), wall_seconds,
                       peak_rss_gb, threads}
    reference_binary_graph3_identical: true      (asserted: the reference's P.graph3 has the restatement's md5 and size)

so that the `-m gpu` full-size tests compare the HIP path with what the reference wrote, not only with a restatement of it.
Build container only (configs[1]: minutes; configs[2]: about an hour and tens of GB).  Usage: python oracle/pin_reference.py c2 [c3 ...]
"""
import ctypes, hashlib, json, os, re, resource, shutil, subprocess, sys, tempfile, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import digests as dg                              # noqa: E402
from make_golden import SynthParams, counters    # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "SAGE2")


def md5_size(path):
    h = hashlib.md5(); n = 0
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 22), b""):
            h.update(blk); n += len(blk)
    return h.hexdigest(), n


def function_seconds(log):
    """'Function buildInitialOverlapGraph() in 27 sec.' lines of the reference's log (hashTable.cpp:126, economyGraph.cpp:488,572 ...)."""
    out = {}
    for m in re.finditer(r"Function\s+(\w+)\(\)\s+in\s+(\d+)\s+sec", open(log, errors="replace").read().replace(",", "")):
        out[m.group(1)] = out.get(m.group(1), 0) + int(m.group(2))
    return out


def pin(name, threads):
    cfg = dg.CONFIGS[name]
    d = json.load(open(dg.path_of(name)))
    lib = ctypes.CDLL(os.path.join(ROOT, "sage2_amd", "libsage2ov.so"))
    lib.sage2ov_synth_write_fasta.argtypes = [ctypes.POINTER(SynthParams), ctypes.c_char_p]
    tmp = tempfile.mkdtemp(prefix="pinref_", dir=os.environ.get("DIGEST_TMP", "/tmp"))
    try:
        fa = os.path.join(tmp, "x.fa")
        p = SynthParams(**cfg["synth"])
        assert lib.sage2ov_synth_write_fasta(ctypes.byref(p), fa.encode()) == 0
        print(f"[{name}] FASTA written: {os.path.getsize(fa) / 1e9:.2f} GB", flush=True)
        env = dict(os.environ, OMP_NUM_THREADS=str(threads), LC_ALL="C")
        t0 = time.time()
        subprocess.run([REF, "-f", fa, "-k", str(cfg["k"]), "-o", os.path.join(tmp, "out"), "-p", "t", "-M", "3"], check=True, env=env,
                       stdout=subprocess.DEVNULL)
        wall = time.time() - t0
        rss = resource.getrusage(resource.RUSAGE_CHILDREN).ru_maxrss / 1e6          # kB -> GB
        os.remove(fa)
        reads, g3, log = (os.path.join(tmp, "out", "t." + e) for e in ("reads", "graph3", "log"))
        gm, gb = md5_size(g3)
        rm, rb = md5_size(reads)
        ref = dict(graph3_md5=gm, graph3_bytes=gb, reads_md5=rm, reads_bytes=rb, counters=counters(log), function_seconds=function_seconds(log),
                   wall_seconds=round(wall, 1), peak_rss_gb=round(rss, 2), threads=threads, command="SAGE2 -f x.fa -k %d -M 3" % cfg["k"])
        print(f"[{name}] reference: {json.dumps(ref)}", flush=True)
        assert (gm, gb) == (d["graph3_md5"], d["graph3_bytes"]), "the reference binary's P.graph3 differs from the restatement's"
        c = ref["counters"]
        for a, b in (("unique_reads", "n_unique"), ("good_reads", "good_reads"), ("contained_extension", "contained_extension"), ("contained_size", "contained_size"),
                     ("left_to_explore", "left_to_explore"), ("edges_inserted", "edges_inserted"), ("transitive_removed", "transitive_removed"),
                     ("long_buckets", "long_buckets")):
            if a in c:
                assert c[a] == d[b], (a, c[a], d[b])
        d["reference_binary"] = ref
        d["reference_binary_graph3_identical"] = True
        json.dump(d, open(dg.path_of(name), "w"), indent=1)
        print(f"[{name}] pinned", flush=True)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    threads = int(os.environ.get("DIGEST_THREADS", "8"))
    for nm in sys.argv[1:]:
        pin(nm, threads)
