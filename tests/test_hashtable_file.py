"""CPU (no GPU needed): P.hashTable -- the slot-by-slot dump of the REFERENCE's double-hashed table (hashTable.cpp:256-273) -- written
by sage2ov_hashtable_save must be the file the reference binary wrote for the same reads (`SAGE2 -M 3 -s` / `-M 2 -s`; md5 + size recorded by
oracle/make_golden.py): table size from the reference's list (reproduced by rule, not copied), start slot, probe step, the 101-entry cap and the
long-bucket flag.  h1 (850 k unique reads) lands in the safe-prime part of the size list."""
import json
import os

import pytest

import fixtures as fx
import sage2_amd as s2


@pytest.mark.parametrize("name", fx.golden_names_all())
def test_hashtable_file_is_the_references(name, tmp_path):
    m = fx.golden(name)
    bases, off = fx.make_reads(m["synth"])
    c = s2.Context(m["k"], device=-2)                      # device-less context: parsing, host organiser, the replay of the serial insertion
    c.reads_add_ascii(bases, off); c.reads_organize()
    p = str(tmp_path / "t.hashTable"); c.hashtable_save(p)
    assert int(open(p).readline()) == m["counters"]["hash_table_size"]
    assert os.path.getsize(p) == m["hashtable_size"] and fx.md5_file(p) == m["hashtable_md5"]
    c.close()


def test_hashtable_file_big_table(tmp_path):
    m = json.load(open(os.path.join(fx.GOLDEN, "h1_c2_1m_k40.hashtable.json")))
    p_ = fx.synth_params(m["synth"])
    c = s2.Context(m["k"], device=-2)
    c.reads_add_synth(p_, s2.synth_genome(p_)); c.reads_organize()
    assert c.reads_stats().unique_reads == m["counters"]["unique_reads"]
    p = str(tmp_path / "t.hashTable"); c.hashtable_save(p)
    assert int(open(p).readline()) == m["counters"]["hash_table_size"] == 6816527
    assert os.path.getsize(p) == m["hashtable_size"] and fx.md5_file(p) == m["hashtable_md5"]
    c.close()


def test_hashtable_file_refuses_tiny_inputs(tmp_path):
    """below 12501 unique reads the reference indexes its size list at [-1] (hashTable.cpp:309-313): no defined file to reproduce"""
    pd = dict(seed=11, genome_len=2000, n_reads=600, read_len=80)
    bases, off = fx.make_reads(pd)
    c = s2.Context(21, device=-2); c.reads_add_ascii(bases, off); c.reads_organize()
    with pytest.raises(s2.Sage2ovError) as e:
        c.hashtable_save(str(tmp_path / "t.hashTable"))
    assert e.value.code == -5
    c.close()
