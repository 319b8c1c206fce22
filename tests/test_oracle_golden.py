"""CPU: the oracle (oracle/sage2_oracle.cpp) must reproduce, byte for byte, the files the REFERENCE binary
wrote for every golden fixture (tests/golden/, made by oracle/make_golden.py).  This is what pins the oracle."""
import pytest

import fixtures as fx
import oracle_lib as ol


@pytest.mark.parametrize("name", fx.golden_names_all())
def test_oracle_matches_reference_files(name, tmp_path):
    m = fx.golden(name)
    bases, off = fx.make_reads(m["synth"])
    o = ol.Oracle(m["k"], threads=4)
    o.add_reads_ascii(bases, off)
    o.organize()
    o.run_all()
    reads_p, g3_p = str(tmp_path / "o.reads"), str(tmp_path / "o.graph3")
    o.write_reads(reads_p)
    o.write_graph3(g3_p)
    assert fx.md5_file(reads_p) == m["reads_md5"]
    assert fx.graph3_matches(g3_p, name)
    c, ref = o.counters(), m["counters"]
    assert c["N"] == ref["unique_reads"] and c["good_reads"] == ref["good_reads"]
    assert c["contained"] == ref["contained_extension"] and c["contained_size"] == ref["contained_size"]
    assert c["edges_inserted"] == ref["edges_inserted"] and c["transitive_removed"] == ref["transitive_removed"]
    assert c["long_buckets"] == ref["long_buckets"] and c["h"] == ref["hash_string_length"]
    o.close()


def test_oracle_thread_count_independent():
    m = fx.golden("g5_mixedlen_k21")
    bases, off = fx.make_reads(m["synth"])
    outs = []
    for t in (1, 4):
        o = ol.Oracle(m["k"], threads=t)
        o.add_reads_ascii(bases, off); o.organize(); o.run_all()
        outs.append((o.export_edges().tobytes(), o.export_initial()[2].tobytes()))
        o.close()
    assert outs[0] == outs[1]


def test_bit_utility_known_answers():
    # SURVEY A.1, derived from utils.cpp:96-122,189-207
    assert ol.pack("ACGT") == bytes([0x1B])
    assert ol.pack("ACG") == bytes([0x18])
    assert ol.pack("TTTTA") == bytes([0xFF, 0x00])
    assert ol.pack("C") == bytes([0x40])
    assert ol.get64(ol.pack("ACGT"), 1, 2) == 0b0110
    assert ol.get64(ol.pack("ACGTACGTACGT"), 2, 8) == int("".join({"A": "00", "C": "01", "G": "10", "T": "11"}[c] for c in "GTACGTAC"), 2)
