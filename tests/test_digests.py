"""The full-size digests (tests/golden/*_digest.json, written by oracle/make_digests.py from the pinned restatement; BASELINE configs[1] and configs[2]
additionally pinned on the reference binary's own run by oracle/pin_reference.py: reference_binary{graph3_md5, reads_md5, counters, seconds}).
CPU: the committed small digests are reproduced by the oracle here (so digest code and oracle cannot drift apart unnoticed).
GPU (-m gpu): the HIP path, through the C ABI, must reproduce every committed digest -- including BASELINE configs[1] (10 M reads),
its noisy variant and configs[2] (50 M reads, the north-star target) -- counters, per-read records, edge list and P.graph3."""
import os

import numpy as np
import pytest

import digests as dg
import fixtures as fx
import oracle_lib as ol
import sage2_amd as s2


@pytest.mark.parametrize("name", ["c1", "c2_1m"])
def test_oracle_reproduces_committed_digest(name, tmp_path):
    want = dg.load(name)
    assert want is not None, f"tests/golden/{name}_digest.json missing: python oracle/make_digests.py {name}"
    cfg = dg.CONFIGS[name]
    assert want["k"] == cfg["k"] and want["synth"] == cfg["synth"]
    bases, off = fx.make_reads(cfg["synth"])
    o = ol.Oracle(cfg["k"], 8); o.add_reads_ascii(bases, off); o.organize(); o.run_all()
    got = {}
    got.update(dg.initial_digest(*o.export_initial()))
    e = o.export_edges(); got.update(dg.edges_digest(e[:, 0], e[:, 1], e[:, 2], e[:, 3], e[:, 4]))
    got.update(dg.reads_digest(*o.export_reads(), cfg["synth"]["read_len"]))
    gp = str(tmp_path / "t.graph3"); o.write_graph3(gp); got.update(dg.file_digest(gp))
    got.update(n_unique=o.counter("N"), edges_inserted=o.counter("edges_inserted"), transitive_removed=o.counter("transitive_removed"))
    assert dg.compare(got, want) == []
    assert want.get("reference_binary_graph3_identical") is True       # make_digests ran the reference binary on this input too
    o.close()


def test_digest_lookup_by_parameters():
    name, d = dg.lookup(21, dict(seed=1, genome_len=200_000, n_reads=100_000, read_len=100, err_ppm=0))
    assert name == "c1" and d["n_unique"] == 78701
    assert dg.lookup(22, dict(seed=1, genome_len=200_000, n_reads=100_000, read_len=100)) == (None, None)


ROUTES = {"library_default": None, "groups_on": "1", "groups_off": "0"}


@pytest.mark.gpu
@pytest.mark.parametrize("name,route", [(n, "library_default") for n in ("c1", "c2_1m", "c2", "c2_noisy", "c2_repeat", "c3")] +
                         [(n, r) for n in ("c2_1m", "c2", "c3") for r in ("groups_on", "groups_off")])
def test_gpu_reproduces_full_size_digest(name, route, tmp_path, monkeypatch):
    """BASELINE sizes the oracle cannot run inside a test: every number the restatement produced for this input (make_digests.py) must
    come out of the device path -- n_unique, N_ov (the numerator of the headline metric), crc32 of connections / extension records /
    status classes / packed reads / canonical edge list, the reduce counters, and the md5 of P.graph3.
    `route`: the access path of the fast kernel's look-ups.  "library_default" is what a user and bench.py get (dev_build_index decides by the data: no
    minimiser groups at BASELINE configs[1] / configs[2]); the two forced routes keep the other half of the look-up code honest at full size."""
    monkeypatch.delenv("SAGE2OV_MINIMIZER_INDEX", raising=False)
    if ROUTES[route] is not None:
        monkeypatch.setenv("SAGE2OV_MINIMIZER_INDEX", ROUTES[route])
    want = dg.load(name)
    if want is None:
        pytest.skip(f"tests/golden/{name}_digest.json not generated")
    cfg = dg.CONFIGS[name]
    p = fx.synth_params(cfg["synth"])
    ctx = s2.Context(cfg["k"], device=0)
    ctx.reads_add_synth(p, s2.synth_genome(p)); ctx.reads_organize(); ctx.run_steps23()
    got = dg.gpu_digest(ctx, cfg["synth"]["read_len"], graph3_path=str(tmp_path / "t.graph3"))
    os.remove(str(tmp_path / "t.graph3"))
    # `keys` on the device = occupied slots: two keys with one 24-bit tag on one probe chain share a slot (about once per 10 M reads; exact,
    # DESIGN.md section 4), so it may fall short of the oracle's distinct-key count by a handful
    assert 0 <= want["keys"] - got["keys"] <= 4 + want["keys"] // 2_000_000, (got["keys"], want["keys"])
    got["keys"] = want["keys"]
    bad = dg.compare(got, want)
    assert bad == [], f"{name}: " + "; ".join(bad)
    if route != "library_default":
        assert ctx.index_stats().minimiser_groups == (1 if route == "groups_on" else 0), "the route asked for is not the route that ran"
    elif name in ("c2", "c3"):
        assert ctx.index_stats().minimiser_groups == 0                    # (what bench.py times: 9.5 % of the reads start a run at 50x coverage, the groups do not pay)
    if name in ("c1", "c2_1m", "c2", "c2_noisy", "c2_repeat", "c3"):      # (round 4: the two inputs that exercise the reduce phase are pinned too -- 8.68 M unresolved reads; 1 178 long buckets)
        # these digests are pinned on the REFERENCE BINARY itself (oracle/make_digests.py for the small ones, oracle/pin_reference.py for BASELINE
        # configs[1] and its noisy / repeat variants (20 minutes each: the reference's serial BFS over 8.7 M unresolved reads is 10 of them) and configs[2]:
        # `SAGE2 -M 3` on the same reads, 46 minutes for the 50 M-read set): the device path's P.graph3 is the file the reference wrote, and so is its P.reads
        assert want.get("reference_binary_graph3_identical") is True
        rb = want.get("reference_binary")
        if rb:
            assert (got["graph3_md5"], got["graph3_bytes"]) == (rb["graph3_md5"], rb["graph3_bytes"])
            if route == "library_default":                  # (step 1 does not depend on the route; 13 GB of text at configs[2])
                rp = str(tmp_path / "t.reads"); ctx.reads_save(rp)
                assert os.path.getsize(rp) == rb["reads_bytes"] and fx.md5_file(rp) == rb["reads_md5"], "P.reads differs from the reference binary's"
                os.remove(rp)
            st_, ost_ = ctx.reads_stats(), ctx.overlap_stats(); rc = rb["counters"]
            assert (st_.unique_reads, st_.good_reads, ost_.contained_extension, ost_.contained_size, ost_.left_to_explore, ost_.edges_inserted, ost_.transitive_removed) == \
                   (rc["unique_reads"], rc["good_reads"], rc["contained_extension"], rc["contained_size"], rc["left_to_explore"], rc["edges_inserted"], rc["transitive_removed"])
            assert ctx.index_stats().long_buckets == rc["long_buckets"]
    # a second pass over the resident reads (atomics-ordered build) gives the same result
    ctx.run_steps23()
    got2 = dg.gpu_digest(ctx, cfg["synth"]["read_len"], with_reads=False); got2["keys"] = want["keys"]
    assert dg.compare(got2, want) == []
    ctx.close()
