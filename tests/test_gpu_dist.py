"""GPU: the multi-rank orchestration (sage2_amd/dist.py::run_steps23_sharded) itself, not only the C-ABI exchange points: `world`
processes share the one GPU of the test box, collectives staged through gloo.  Mixed read lengths (containment marks that one rank's
probe places on reads of another rank, economyGraph.cpp:735), noisy data (reduce phase) and long buckets (ranked reduce)."""
import os
import subprocess
import sys

import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("minimiser_groups_on")]      # (small inputs: the groups' half of the look-up code is exercised by request, conftest.py)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("name,world", [("g5_mixedlen_k21", 2), ("g5_mixedlen_k21", 3), ("g3_noisy_rep_k21", 2), ("g4_highcopy_k21", 3)])
def test_run_steps23_sharded_matches_reference(name, world, tmp_path):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(29531 + world), os.path.join(ROOT, "tests", "dist_gpu_worker.py"), name, str(tmp_path)]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env, timeout=600)
    out = r.stdout.decode()
    assert r.returncode == 0 and f"DIST_GPU_OK {world}" in out, out[-3000:]


@pytest.mark.parametrize("name", ["g5_mixedlen_k21", "g3_noisy_rep_k21", "g4_highcopy_k21"])
def test_run_steps23_sharded_on_rccl_one_rank(name, tmp_path):
    """backend "nccl" (= RCCL): the device-tensor branches of sage2_amd/dist.py -- all-gather of records and edge buckets, MAX all-reduce of the
    containment planes -- execute on RCCL itself, with the one rank a one-GPU box allows.  Same files as the reference's."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", "29541", os.path.join(ROOT, "tests", "dist_gpu_worker.py"), name, str(tmp_path), "nccl"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env, timeout=600)
    out = r.stdout.decode()
    assert r.returncode == 0 and "DIST_GPU_OK 1 nccl" in out, out[-3000:]


def test_bench_sharded_path_on_rccl_one_rank():
    """bench.py under torchrun with one rank and the multi-rank path forced (process group on RCCL, broadcast of the read store, the three
    exchanges): its edge-list crc must equal the single-context run's, and the JSON must say what ran."""
    import json
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", SAGE2OV_BENCH_FORCE_SHARDED="1")
    common = ["--steps", "1", "--warmup", "1", "--reads", "200000", "--no-cpu-baseline", "--no-noisy-variant", "--no-c2", "--no-step4"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1", "--master-port", "29543",
           os.path.join(ROOT, "bench.py"), "--gpus", "1"] + common
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    a = json.loads([ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")][-1])
    assert a["collectives"] == "nccl" and a["world_size"] == 1 and a["config"]["partition"].startswith("locality-order position range")
    env1 = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + common, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env1, timeout=900)
    assert r1.returncode == 0, r1.stderr.decode()[-3000:]
    b = json.loads([ln for ln in r1.stdout.decode().splitlines() if ln.startswith("{")][-1])
    assert b["collectives"] is None and a["config"]["edges_crc32"] == b["config"]["edges_crc32"] and a["config"]["verified_overlaps"] == b["config"]["verified_overlaps"]


def test_bench_starts_its_own_ranks_without_a_launcher():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment: bench.py itself starts one child per rank (the parent never touches the
    GPU) and relays rank 0's line, which must say two ranks ran.  Both ranks share the one GPU here, so the collectives are staged through gloo."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(SAGE2OV_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--reads", "200000",
                        "--no-cpu-baseline", "--no-noisy-variant", "--no-c2", "--no-step4"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    a = json.loads([ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")][-1])
    assert a["n_gpus"] == 2 and a["world_size"] == 2 and a["collectives"] == "gloo"


@pytest.mark.parametrize("name,gpus,flags", [("g5_mixedlen_k21", 2, ["--share-gpu"]), ("g3_noisy_rep_k21", 3, ["--share-gpu"]), ("g4_highcopy_k21", 2, ["--share-gpu"]),
                                             ("g2_clean150_k40", 1, ["--force-multi"]), ("g3_noisy_rep_k21", 1, ["--force-multi"])])
def test_cli_multi_gpu_mode_writes_reference_files(name, gpus, flags, tmp_path):
    """`sage2ov --gpus G`: the C++ host driver of the multi-GPU path (sage2_amd/csrc/sage2ov_multi.cpp: one thread and one context per rank, the four
    exchanges).  With --share-gpu all ranks run on the one GPU of the test box and exchange by device copies; with --force-multi one rank goes through
    RCCL itself (ncclCommInitAll, ncclAllGather, ncclAllReduce).  P.reads and P.graph3 must be the reference's files either way."""
    import fixtures as fx, sage2_amd as s2
    m = fx.golden(name)
    fa = str(tmp_path / "x.fa"); s2.synth_write_fasta(fx.synth_params(m["synth"]), fa)
    out = str(tmp_path / "out")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([os.path.join(ROOT, "sage2_amd", "sage2ov"), "-f", fa, "-k", str(m["k"]), "-o", out, "-p", "t", "-M", "3", "-G", str(gpus)] + flags,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env, timeout=600)
    assert r.returncode == 0, r.stdout.decode()[-2000:]
    assert fx.md5_file(os.path.join(out, "t.reads")) == m["reads_md5"]
    assert fx.graph3_matches(os.path.join(out, "t.graph3"), name)
    log = open(os.path.join(out, "t.log")).read()
    assert f"STEPS 2-3 on {gpus}" in log and ("(RCCL)" in log) == ("--force-multi" in flags)
    for key, lab in (("contained_extension", "Total contained by extension"), ("transitive_removed", "Transitive edge removed")):
        if key in m["counters"]:
            assert f"{lab}: {m['counters'][key]}" in log


@pytest.mark.parametrize("gpus,flags,fail_rank", [(3, ["--share-gpu"], 1), (2, ["--share-gpu"], 0), (1, ["--force-multi"], 0)])
def test_cli_multi_gpu_run_with_a_failed_rank_ends_non_zero(gpus, flags, fail_rank, tmp_path):
    """A rank that fails (here: injected before its first step, `--fail-rank`) must take the run down -- exit status 1, the failing rank named -- instead of
    leaving its peers inside a collective or at the thread barrier for ever (sage2ov_multi.cpp; both transports)."""
    import fixtures as fx, sage2_amd as s2
    m = fx.golden("g2_clean150_k40")
    fa = str(tmp_path / "x.fa"); s2.synth_write_fasta(fx.synth_params(m["synth"]), fa)
    r = subprocess.run([os.path.join(ROOT, "sage2_amd", "sage2ov"), "-f", fa, "-k", str(m["k"]), "-o", str(tmp_path / "out"), "-p", "t", "-M", "3", "-G", str(gpus),
                        "--fail-rank", str(fail_rank)] + flags, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"), timeout=300)
    out = r.stdout.decode()
    assert r.returncode not in (0, None) and f"rank {fail_rank}" in out and "failure injected" in out, (r.returncode, out[-2000:])
    assert not os.path.exists(str(tmp_path / "out" / "t.graph3"))


@pytest.fixture(scope="module")
def configs3_fasta(tmp_path_factory):
    """BASELINE configs[3]'s reads (= configs[2]'s: 50 M x 150 bp, 150 Mb genome, seed 3) as the FASTA file `sage2ov -f` takes (8.2 GB)"""
    import digests as dg, fixtures as fx, sage2_amd as s2
    d = tmp_path_factory.mktemp("configs3")
    fa = str(d / "c3.fa"); s2.synth_write_fasta(fx.synth_params(dg.CONFIGS["c3"]["synth"]), fa)
    yield fa
    os.remove(fa)


@pytest.mark.parametrize("ranks", [2, 8])
def test_configs3_workload_through_the_sharded_path(ranks, configs3_fasta, tmp_path, monkeypatch):
    """BASELINE configs[3]: the 50 M-read workload (42.5 M unique reads) through the multi-rank code path -- `sage2ov -G ranks`: the C++ driver, one thread and one
    context per rank, reads and index replicated, the probe pass cut by position range, the four exchanges -- with 2 and 8 ranks sharing the one GPU of the
    test box (memory-diet mode: eight replicas of a 26 GB context do not fit 288 GB with room to spare).  P.graph3 and P.reads must be the files the REFERENCE
    BINARY wrote for this input (tests/golden/c3_digest.json: reference_binary), the log's counters the reference's."""
    import digests as dg, fixtures as fx
    monkeypatch.delenv("SAGE2OV_MINIMIZER_INDEX", raising=False)              # (the library's own route: what a multi-GPU user gets)
    want = dg.load("c3"); rb = want["reference_binary"]
    out = str(tmp_path / "out")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", SAGE2OV_MEMORY_DIET="1")
    r = subprocess.run([os.path.join(ROOT, "sage2_amd", "sage2ov"), "-f", configs3_fasta, "-k", "40", "-o", out, "-p", "t", "-M", "3", "-G", str(ranks), "--share-gpu"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env, timeout=1500)
    assert r.returncode == 0, r.stdout.decode()[-3000:]
    g3, rd = os.path.join(out, "t.graph3"), os.path.join(out, "t.reads")
    assert (os.path.getsize(g3), fx.md5_file(g3)) == (rb["graph3_bytes"], rb["graph3_md5"]), "P.graph3 of the sharded run differs from the reference binary's"
    os.remove(g3)
    assert (os.path.getsize(rd), fx.md5_file(rd)) == (rb["reads_bytes"], rb["reads_md5"]), "P.reads differs from the reference binary's"
    os.remove(rd)
    log = open(os.path.join(out, "t.log")).read()
    assert f"STEPS 2-3 on {ranks} ranks sharing GPU" in log
    for key, lab in (("contained_extension", "Total contained by extension"), ("contained_size", "Total contained by size"), ("left_to_explore", "Total left to explore"),
                     ("edges_inserted", "Total edges inserted"), ("transitive_removed", "Transitive edge removed")):
        assert f"{lab}: {rb['counters'][key]}" in log, (lab, rb["counters"][key])
    assert f"Verified overlaps: {want['n_ov']}" in log and f"Edges in the graph: {want['edges']}" in log
