"""Shared helpers: golden fixture metadata, synthetic reads, md5."""
import glob
import gzip
import hashlib
import json
import os

import sage2_amd as s2

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def golden_names():
    return sorted(os.path.basename(p)[:-5] for p in glob.glob(os.path.join(GOLDEN, "*.json")))


def golden(name):
    return json.load(open(os.path.join(GOLDEN, name + ".json")))


def golden_graph3(name) -> bytes:
    return gzip.open(os.path.join(GOLDEN, name + ".graph3.gz"), "rb").read()


def md5_file(path):
    h = hashlib.md5()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def synth_params(d) -> s2.SynthParams:
    return s2.SynthParams(**d)


def make_reads(pd):
    """(bases u8 array, offsets u64 array) of the synthetic data set described by dict pd."""
    p = synth_params(pd)
    g = s2.synth_genome(p)
    return s2.synth_reads_ascii(p, g)
