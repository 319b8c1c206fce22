#!/usr/bin/env python3
"""oracle/make_golden.py -- TEST INFRASTRUCTURE: regenerate tests/golden/ with the REFERENCE binary.

Runs in the build container only (needs oracle/_ref/SAGE2, built from /root/reference by
oracle/Makefile).  For every fixture it writes the FASTA with the repo's own deterministic
generator (sage2ov_synth_write_fasta), runs `SAGE2 -f x.fa -k K -o out -p t -M 3 -s`
(main.cpp:37-132) and stores
    tests/golden/<name>.json       generator parameters, k, md5/size of t.reads and t.graph3,
                                   the reference's log counters
    tests/golden/<name>.graph3.gz  the reference's t.graph3, verbatim
Fixtures are data (inputs = generator parameters, outputs = reference files); no reference
source is stored.  Usage: python oracle/make_golden.py [libpath-with-synth]
"""
import ctypes, gzip, hashlib, json, os, re, shutil, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "SAGE2")

class SynthParams(ctypes.Structure):
    _fields_ = [("seed", ctypes.c_uint64), ("genome_len", ctypes.c_uint64), ("n_reads", ctypes.c_uint64),
                ("read_len", ctypes.c_uint32), ("read_len_min", ctypes.c_uint32), ("err_ppm", ctypes.c_uint32),
                ("n_repeat_families", ctypes.c_uint32), ("repeat_copies", ctypes.c_uint32), ("repeat_len", ctypes.c_uint32)]

FIXTURES = {
    # name: (k, threads, params)
    "g1_clean100_k21":   (21, 8, dict(seed=1, genome_len=40000, n_reads=20000, read_len=100)),
    "g2_clean150_k40":   (40, 8, dict(seed=2, genome_len=60000, n_reads=20000, read_len=150)),
    "g3_noisy_rep_k21":  (21, 8, dict(seed=3, genome_len=40000, n_reads=20000, read_len=100, err_ppm=3000,
                                      n_repeat_families=4, repeat_copies=5, repeat_len=500)),
    "g4_highcopy_k21":   (21, 8, dict(seed=4, genome_len=120000, n_reads=40000, read_len=100, err_ppm=500,
                                      n_repeat_families=1, repeat_copies=500, repeat_len=150)),
    "g5_mixedlen_k21":   (21, 1, dict(seed=5, genome_len=40000, n_reads=24000, read_len=100, read_len_min=70, err_ppm=2000)),
    "g6_k70_150":        (70, 8, dict(seed=6, genome_len=60000, n_reads=20000, read_len=150, err_ppm=1000)),
    # 250-bp reads with errors and a few repeats: 8-word slots with bases and length sharing the last dword, 16-dword compares
    "g8_noisy250_k45":   (45, 8, dict(seed=8, genome_len=90000, n_reads=18000, read_len=250, err_ppm=1500,
                                      n_repeat_families=2, repeat_copies=6, repeat_len=400)),
    # a bigger set with several high-copy repeat families and read errors: hundreds of long buckets (hashTable.cpp:111-123), one-sided
    # discovery, i.e. the reduce phase's result depends on the reference's serial exploration order (economyGraph.cpp:513-564) at a size
    # where the device takes its ranked path by default.  Too big to commit as a file: md5 + size + counters only.
    "g9_repeats160k_k40": (40, 8, dict(seed=9, genome_len=500000, n_reads=160000, read_len=150, err_ppm=1500,
                                       n_repeat_families=6, repeat_copies=250, repeat_len=350)),
    # reads of 520..900 bases with errors: the 32-word slot layout (505 .. 1018 bases, 11-bit length field), sequential probe kernel only
    "g10_long900_k55":   (55, 8, dict(seed=10, genome_len=150000, n_reads=16000, read_len=900, read_len_min=520, err_ppm=800)),
    # hand-made input (tests/fixtures.py::recipe_reads): palindromic region, tandem repeat, mirrored duplicates
    "g7_palindrome_tandem_k21": (21, 8, dict(recipe="palindrome_tandem", seed=7, half=700, flank=24000, tandem_units=60, read_len=100, step=3)),
}

def md5(path):
    h = hashlib.md5()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()

def counters(log):
    txt = open(log, errors="replace").read().replace(",", "")
    pats = {"unique_reads": r"Number of unique reads:\s*(\d+)", "good_reads": r"Good reads:\s*(\d+)|Good reads in file:\s*(\d+)",
            "contained_extension": r"Total contained by extension:\s*(\d+)", "contained_size": r"Total contained by size:\s*(\d+)",
            "left_to_explore": r"Total left to explore:\s*(\d+)", "edges_inserted": r"Total edges inserted:\s*(\d+)",
            "transitive_removed": r"Transitive edge removed:\s*(\d+)", "long_buckets": r"hash elements over threshold:\s*(\d+)",
            "hash_string_length": r"Hash string length:\s*(\d+)", "hash_table_size": r"Hash table size:\s*(\d+)"}
    out = {}
    for k, p in pats.items():
        m = re.search(p, txt)
        if m: out[k] = int([g for g in m.groups() if g][0])
    return out

# P.hashTable only (`SAGE2 -M 2 -s`): a read set big enough for the safe-prime part of the reference's table-size list (8N > 1 000 003)
HASHTABLE_ONLY = {
    "h1_c2_1m_k40": (40, 8, dict(seed=2, genome_len=3_000_000, n_reads=1_000_000, read_len=150)),
}

def make_hashtable_only(lib, gold):
    for name, (k, threads, pd) in HASHTABLE_ONLY.items():
        tmp = tempfile.mkdtemp(prefix="sage2gold_")
        try:
            fa = os.path.join(tmp, "x.fa"); p = SynthParams(**pd)
            assert lib.sage2ov_synth_write_fasta(ctypes.byref(p), fa.encode()) == 0
            env = dict(os.environ, OMP_NUM_THREADS=str(threads), LC_ALL="C")
            subprocess.run([REF, "-f", fa, "-k", str(k), "-o", os.path.join(tmp, "out"), "-p", "t", "-M", "2", "-s"], check=True, env=env, stdout=subprocess.DEVNULL)
            ht, reads, log = (os.path.join(tmp, "out", "t." + e) for e in ("hashTable", "reads", "log"))
            meta = dict(name=name, k=k, ref_threads=threads, synth=pd, reads_md5=md5(reads), hashtable_md5=md5(ht), hashtable_size=os.path.getsize(ht), counters=counters(log))
            json.dump(meta, open(os.path.join(gold, name + ".hashtable.json"), "w"), indent=1, sort_keys=True)
            print(name, meta["counters"], "hashTable", meta["hashtable_size"])
        finally:
            shutil.rmtree(tmp, ignore_errors=True)

def main():
    lib = ctypes.CDLL(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "sage2_amd", "libsage2ov.so"))
    lib.sage2ov_synth_write_fasta.argtypes = [ctypes.POINTER(SynthParams), ctypes.c_char_p]
    gold = os.path.join(ROOT, "tests", "golden"); os.makedirs(gold, exist_ok=True)
    only = set(sys.argv[2:])
    if not only or "hashtable" in only: make_hashtable_only(lib, gold)
    for name, (k, threads, pd) in FIXTURES.items():
        if only and name not in only: continue
        tmp = tempfile.mkdtemp(prefix="sage2gold_")
        try:
            fa = os.path.join(tmp, "x.fa")
            if "recipe" in pd:
                sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
                import fixtures
                fixtures.write_recipe_fasta(pd, fa)
            else:
                p = SynthParams(**pd)
                assert lib.sage2ov_synth_write_fasta(ctypes.byref(p), fa.encode()) == 0
            env = dict(os.environ, OMP_NUM_THREADS=str(threads), LC_ALL="C")
            subprocess.run([REF, "-f", fa, "-k", str(k), "-o", os.path.join(tmp, "out"), "-p", "t", "-M", "3", "-s"], check=True, env=env,
                           stdout=subprocess.DEVNULL)
            reads, g3, log = (os.path.join(tmp, "out", "t." + e) for e in ("reads", "graph3", "log"))
            meta = dict(name=name, k=k, ref_threads=threads, synth=pd, fasta_md5=md5(fa),
                        reads_md5=md5(reads), reads_size=os.path.getsize(reads),
                        graph3_md5=md5(g3), graph3_size=os.path.getsize(g3), counters=counters(log))
            ht = os.path.join(tmp, "out", "t.hashTable")                     # (-s: the reference also dumped its table, hashTable.cpp:256)
            meta["hashtable_md5"], meta["hashtable_size"] = md5(ht), os.path.getsize(ht)
            if os.path.getsize(g3) <= 4 << 20:
                with open(g3, "rb") as fi, gzip.GzipFile(os.path.join(gold, name + ".graph3.gz"), "wb", mtime=0) as fo:
                    shutil.copyfileobj(fi, fo)
            else:
                meta["graph3_file"] = "not committed (size): compare md5 and size"
            json.dump(meta, open(os.path.join(gold, name + ".json"), "w"), indent=1, sort_keys=True)
            print(name, meta["counters"], "graph3", meta["graph3_size"])
        finally:
            shutil.rmtree(tmp, ignore_errors=True)

if __name__ == "__main__":
    main()
