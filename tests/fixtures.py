"""Shared helpers: golden fixture metadata, synthetic reads, md5."""
import glob
import gzip
import hashlib
import json
import os

import sage2_amd as s2

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def golden_names_all():
    """every fixture the reference binary produced (oracle/make_golden.py), including the ones pinned by md5 + size only"""
    return sorted(os.path.basename(p)[:-5] for p in glob.glob(os.path.join(GOLDEN, "*.json")) if not p.endswith(".step4.json") and not p.endswith("_digest.json") and not p.endswith(".hashtable.json"))


def golden_names():
    """the fixtures whose P.graph3 is committed (and that have a step-4 dump)"""
    return [n for n in golden_names_all() if os.path.exists(os.path.join(GOLDEN, n + ".graph3.gz"))]


def golden(name):
    return json.load(open(os.path.join(GOLDEN, name + ".json")))


def golden_graph3(name) -> bytes:
    return gzip.open(os.path.join(GOLDEN, name + ".graph3.gz"), "rb").read()


def graph3_matches(path, name):
    """the written P.graph3 against the reference's: byte for byte where the file is committed, md5 + size for the big fixtures"""
    gz = os.path.join(GOLDEN, name + ".graph3.gz")
    if os.path.exists(gz):
        return open(path, "rb").read() == gzip.open(gz, "rb").read()
    m = golden(name)
    return os.path.getsize(path) == m["graph3_size"] and md5_file(path) == m["graph3_md5"]


def md5_file(path):
    h = hashlib.md5()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def synth_params(d) -> s2.SynthParams:
    return s2.SynthParams(**d)


def recipe_reads(pd):
    """Hand-made inputs the generator never produces (dict with a "recipe" key) -> list of read strings.
    palindrome_tandem: a genome that is its own reverse complement around a centre (reads at mirrored positions share a canonical
    form; reads across the centre equal their own reverse complement), a tandem repeat of period 7 (several overlaps per read pair,
    a read's prefix equal to its own later windows), embedded in random sequence long enough for the reference (> 12.5 k unique reads)."""
    import numpy as np
    assert pd["recipe"] == "palindrome_tandem"
    rng = np.random.default_rng(pd["seed"])
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    def rc(s): return "".join(comp[c] for c in reversed(s))
    def rnd(n): return "".join(rng.choice(list("ACGT"), size=n))
    half = rnd(pd["half"])
    genome = rnd(pd["flank"]) + half + rc(half) + rnd(200) + "ACGTTGA" * pd["tandem_units"] + rnd(pd["flank"])
    L, step, reads = pd["read_len"], pd["step"], []
    for i, p0 in enumerate(range(0, len(genome) - L + 1, step)):
        s = genome[p0:p0 + L]
        reads.append(s if i % 2 == 0 else rc(s))
    return reads


def write_recipe_fasta(pd, path):
    with open(path, "w") as f:
        for i, r in enumerate(recipe_reads(pd)):
            f.write(">r%d\n%s\n" % (i, r))


def make_reads(pd):
    """(bases u8 array, offsets u64 array) of the synthetic data set described by dict pd."""
    if "recipe" in pd:
        import numpy as np
        reads = recipe_reads(pd)
        bases = np.frombuffer("".join(reads).encode(), dtype=np.uint8).copy()
        off = np.zeros(len(reads) + 1, dtype=np.uint64); off[1:] = np.cumsum([len(r) for r in reads])
        return bases, off
    p = synth_params(pd)
    g = s2.synth_genome(p)
    return s2.synth_reads_ascii(p, g)
