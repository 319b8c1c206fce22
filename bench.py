#!/usr/bin/env python3
"""bench.py -- overlaps/sec of the MI355X-native SAGE2 read-overlap path (steps 2-3) on synthetic paired-end reads.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N>1 launched through
torch.distributed.run, one rank per GPU.  One "step" = one pass of the hot path over the resident read set:
index build + initial pass (probe/verify/extension + reciprocal) + reduce + sort/convert, i.e. the timed
region T of SURVEY.md 8(d) ("packed unique reads resident in HBM" -> "canonical edge list resident").
value = N_ov / T, N_ov = verified suffix-prefix overlaps (sum of the reference's `connections`).
Workload at N=1: BASELINE.json configs[2], the north-star target (50 M x 150 bp, k=40, 150 Mb genome, seed 3: SURVEY 8d); configs[1]
(10 M reads, seed 2) is timed beside it as `c2`.  N>1: the same read set (configs[3]), read ids range-partitioned over the ranks
(strong scaling), records/flags/edge buckets exchanged over RCCL.  The result of every workload that has an oracle-generated digest
(tests/golden/*_digest.json: N_ov, crc32 of the edge list, ...) is asserted against it before anything is printed.

Adds to the JSON line: "roofline" (dominant kernel = k_probe, HIP-event timed inside the library) and
"cpu_baseline" (the reference itself, oracle/_ref, on a bounded sample of the same workload, rank 0, N=1 only).
"""
import argparse
import ctypes as C
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))


def csrc_sha1():
    """sha1 over the whole device side of the library -- sage2ov_device.hip (launchers: grids, chunk sizes, which access path), every kernels_*.inc and the
    internal header -- in name order.  The recorded PMC figures under profiles/ (probe_traffic.json, probe_insts.json, index_traffic.json) carry the hash of the
    source they were measured on; bench.py reports them only while the source is the same, null otherwise (round 3 keyed them on two files: a changed launch
    shape kept the old figure alive)."""
    import glob, hashlib
    d = os.path.join(ROOT, "sage2_amd", "csrc")
    names = sorted(glob.glob(os.path.join(d, "kernels_*.inc")) + [os.path.join(d, "sage2ov_device.hip"), os.path.join(d, "sage2ov_internal.h")])
    h = hashlib.sha1()
    for nm in names:
        h.update(os.path.basename(nm).encode()); h.update(open(nm, "rb").read())
    return h.hexdigest()
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
SLOT_ALG = 16                  # algorithmic index slot = 8-B key + 8-B payload


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def algorithmic_bytes(n_unique, n_ov, n_edges, L, k):
    """A_total = N(2B + 4S + W S + 16) + N_ov B + 16 N_e; the probe kernel's share is
    N(B + W S + 16) + N_ov B (read the read once, one slot per window, one neighbour per overlap, two ext records)."""
    B = (L + 3) // 4
    h = min(k, 64)
    W = L - h + 1
    total = n_unique * (2 * B + 4 * SLOT_ALG + W * SLOT_ALG + 16) + n_ov * B + 16 * n_edges
    probe = n_unique * (B + W * SLOT_ALG + 16) + n_ov * B
    return total, probe


XGMI_LINK_GBS = 153.0          # MI355X_MICROARCH.md / task notes: 7 xGMI links x ~153 GB/s per GPU (both directions of a link together)


def scaling_model(n_unique, n_edges, ms1, ph, alt=None):
    """MODEL, UNMEASURED: the step time on G ranks predicted from the ONE-GPU phase times of this run (no multi-GPU hardware is available to the
    builder; the driver's scaling runs are the measurement).  Per rank: replicated = index build (no minimiser groups: the library decides by the share of reads without a predecessor, 9.5 % at this coverage: dev_build_index) + cond half of the reciprocal pass + hit lists / adjacency of the reduce phase + convert; sharded = probe pass,
    emit half of the reciprocal pass, marks of the reduce phase; exchanges = records (16 B/read), containment planes (2 B/read, all-reduce), edge
    and survivor buckets (16 B/edge), each rank receiving (G-1)/G of the bytes over G-1 links at `link_efficiency` of one direction of a link,
    plus a fixed latency per collective.  ph: phases of the one-GPU step (ms); alt: index / probe times measured without minimiser groups."""
    eff, lat_ms, ncoll = 0.7, 0.05, 8
    out = {"label": "model, unmeasured", "assumptions": {"xgmi_link_GBs": XGMI_LINK_GBS, "link_efficiency": eff, "receive_GBs_per_peer": XGMI_LINK_GBS / 2 * eff,
                                                         "collectives_per_step": ncoll, "latency_ms_per_collective": lat_ms,
                                                         "replicated": "index build, cond half of the reciprocal pass, hit lists + adjacency of the reduce phase, convert",
                                                         "sharded": "probe pass, emit half of the reciprocal pass, marks of the reduce phase"},
           "one_gpu_ms": ms1, "ranks": {}}
    for G in (2, 4, 8):
        groups = False
        index = ph["index_ms"] if (groups or not alt) else alt["index_ms"]
        probe = (ph["probe_ms"] if (groups or not alt) else alt["probe_ms"]) / G
        cond = ph.get("reciprocal_cond_ms", ph["reciprocal_ms"])
        recip = cond + (ph["reciprocal_ms"] - cond) / G
        marks = ph.get("reduce_marks_ms", 0.0)
        reduce_ = (ph["reduce_ms"] - marks) + marks / G
        bytes_recv = (16.0 * n_unique + 2 * 2.0 * n_unique + 16.0 * n_edges) * (G - 1) / G
        exch = bytes_recv / ((G - 1) * XGMI_LINK_GBS / 2 * eff * 1e9) * 1e3 + ncoll * lat_ms
        t = index + probe + recip + reduce_ + ph["convert_ms"] + exch
        out["ranks"][str(G)] = {"ms": t, "speedup": ms1 / t, "index_ms": index, "probe_ms": probe, "reciprocal_ms": recip, "reduce_ms": reduce_, "convert_ms": ph["convert_ms"],
                                "exchange_ms": exch, "minimiser_groups": bool(groups)}
    return out


def cpu_baseline(args, s2, fx):
    """Time the REFERENCE (oracle/_ref/libsage2ref_driver.so: the reference's own classes, built from
    /root/reference by oracle/Makefile) on a bounded sample of the same workload, and diff its P.graph3
    against ours on that sample.  Falls back to the CPU restatement (kind "port") when _ref is absent."""
    n = args.cpu_sample_reads
    cov = args.reads * args.read_len / args.genome
    pd = dict(seed=args.seed + 1000, genome_len=int(n * args.read_len / cov), n_reads=n, read_len=args.read_len)
    if args.cpu_full:       # BASELINE.json configs[1] itself, every read of it (the generator parameters of the c2 digest): minutes of CPU, not part of the default run
        n = 10_000_000
        pd = dict(seed=2, genome_len=3 * n, n_reads=n, read_len=args.read_len)
    p = fx.synth_params(pd)
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        cores = os.cpu_count() or 1
    # the GPU box gives one GPU's job a 16-core share of the host (more OpenMP threads only oversubscribe it)
    threads = args.cpu_threads if args.cpu_threads else min(cores, 16)
    sample = f"{n} x {args.read_len} bp reads, k={args.k}, {pd['genome_len']} bp genome, seed {pd['seed']} " + ("(BASELINE.json configs[1], the whole input: --cpu-full)" if args.cpu_full else "(same coverage as the GPU workload)")
    tmp = tempfile.mkdtemp(prefix="sage2bench_")
    out = None
    try:
        # our result on the sample: N_ov and the file to diff against
        ctx = s2.Context(args.k, device=0)
        g = s2.synth_genome(p)
        ctx.reads_add_synth(p, g)
        ctx.reads_organize()
        ctx.run_steps23()
        nov = ctx.overlap_stats().verified_overlaps
        ours = os.path.join(tmp, "ours.graph3")
        ctx.graph_save(ours)
        ours4, ours4_ms = os.path.join(tmp, "ours.graph4"), None
        try:
            ctx.graph_simplify(); ctx.graph4_save(ours4); ours4_ms = ctx.simplify_stats().device_ms
        except Exception as e:      # noqa
            log("[bench] step 4 on the sample failed:", repr(e)); ours4 = None
        ctx.close()
        drv = os.path.join(ROOT, "oracle", "_ref", "libsage2ref_driver.so")
        if os.path.exists(drv) and not args.cpu_port:
            fa = os.path.join(tmp, "sample.fa")
            s2.synth_write_fasta(p, fa)
            L = C.CDLL(drv)
            t = (C.c_double * 6)()
            c = (C.c_ulonglong * 3)()
            pref = os.path.join(tmp, "ref")
            rc = L.sage2ref_run_steps123(fa.encode(), args.k, threads, pref.encode(), t, c)
            assert rc == 0
            T = t[2] + t[3] + t[4] + t[5]
            same = open(pref + ".graph3", "rb").read() == open(ours, "rb").read()
            out = dict(value=nov / T, unit="overlaps/s", cores=threads, kind="reference", sample=sample,
                       seconds=dict(index=t[2], initial=t[3], reduce=t[4], sort_convert=t[5]),
                       graph3_identical_to_gpu=bool(same))
            # the same path on ONE thread on the SAME sample: the reference's cost per overlap grows with the size of its table, so
            # only equal samples compare -- and on this host its OpenMP loop is slower than its serial one (per-k-mer mallocs, utils.cpp:171-207)
            t1 = (C.c_double * 6)(); cc1 = (C.c_ulonglong * 3)()
            if not args.cpu_full and L.sage2ref_run_steps123(fa.encode(), args.k, 1, None, t1, cc1) == 0:
                T1 = t1[2] + t1[3] + t1[4] + t1[5]
                out["one_thread"] = dict(value=nov / T1, unit="overlaps/s", cores=1, sample="the same sample", seconds=dict(index=t1[2], initial=t1[3], reduce=t1[4], sort_convert=t1[5]))
            if hasattr(L, "sage2ref_run_step4") and ours4:
                # step 4 of the reference on the same sample (its loaders, its loop, its writer), against our P.graph4
                t4 = (C.c_double * 2)(); c4 = (C.c_ulonglong * 4)()
                L.sage2ref_run_step4.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_ulonglong)]
                if L.sage2ref_run_step4(pref.encode(), args.k, threads, (pref + ".graph4").encode(), t4, c4) == 0:
                    out["step4"] = dict(seconds=t4[1], load_seconds=t4[0], nodes_contracted=int(c4[2]), removed=int(c4[3]), gpu_device_ms=ours4_ms,
                                        graph4_identical_to_gpu=bool(open(pref + ".graph4", "rb").read() == open(ours4, "rb").read()))
        else:
            import oracle_lib as ol
            bases, off = s2.synth_reads_ascii(p, g)
            o = ol.Oracle(args.k, threads)
            o.add_reads_ascii(bases, off)
            o.organize()
            o.run_all()
            T = sum(o.time(i) for i in range(4))
            gp = os.path.join(tmp, "port.graph3")
            o.write_graph3(gp)
            same = open(gp, "rb").read() == open(ours, "rb").read()
            out = dict(value=o.counter("n_ov") / T, unit="overlaps/s", cores=threads, kind="port", sample=sample,
                       seconds=dict(index=o.time(0), initial=o.time(1), reduce=o.time(2), sort_convert=o.time(3)),
                       graph3_identical_to_gpu=bool(same))
            o.close()
    finally:
        import shutil
        shutil.rmtree(tmp, ignore_errors=True)
    return out


def cpu_baseline_subprocess(args):
    """Run cpu_baseline() in a fresh interpreter (no torch, its own OpenMP runtime): the reference's objects and
    torch's bundled runtimes are kept apart, and a crash in the CPU leg cannot take the bench line with it."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", "--reads", str(args.reads), "--read-len", str(args.read_len),
           "--k", str(args.k), "--genome", str(args.genome), "--seed", str(args.seed), "--cpu-sample-reads", str(args.cpu_sample_reads),
           "--cpu-threads", str(args.cpu_threads)] + (["--cpu-port"] if args.cpu_port else []) + (["--cpu-full"] if args.cpu_full else [])
    try:
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=3000 if args.cpu_full else 1500)
        for line in reversed(r.stdout.decode().splitlines()):
            if line.startswith("{"):
                return json.loads(line)
        log("[bench] cpu baseline worker failed:", r.returncode, r.stderr.decode()[-2000:])
    except Exception as e:      # noqa
        log("[bench] cpu baseline worker failed:", repr(e))
    return None


def spawn_ranks(n):
    """One child process per rank (fresh interpreters; the parent initialises no GPU runtime, so nothing is ever exec'ed over a live HIP
    context), rendezvous on 127.0.0.1.  Returns the first non-zero exit code; a rank that fails takes the others down with it."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0)); port = so.getsockname()[1]
    base = dict(os.environ, WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = []
    for r in range(n):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # rank 0's line is drained by a reader thread while ALL children are polled: a rank >= 1 that dies leaves rank 0 inside a collective with its stdout open, and a
    # parent blocked in procs[0].stdout.read() would never get to the terminate loop
    import threading
    buf = []
    rd = threading.Thread(target=lambda: buf.append(procs[0].stdout.read()), daemon=True); rd.start()
    rc = 0
    while any(pr.poll() is None for pr in procs):
        for r, pr in enumerate(procs):
            c = pr.poll()
            if c and not rc:
                rc = c
                log(f"[bench] rank {r} exited with code {c}")
                for q in procs:                              # the exact children started above, nothing else
                    if q.poll() is None:
                        q.terminate()
        time.sleep(0.05)
    for r, pr in enumerate(procs):
        c = pr.wait()
        if c and not rc:
            rc = c; log(f"[bench] rank {r} exited with code {c}")
    rd.join(timeout=10)
    out = (buf[0] if buf else b"").decode()
    sys.stdout.write(out); sys.stdout.flush()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-step4", action="store_true", help="skip the step-4 (graph simplification) figures reported beside the metric")
    ap.add_argument("--reads", type=int, default=50_000_000)      # BASELINE.json configs[2]: the target configuration
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--k", type=int, default=40)
    ap.add_argument("--genome", type=int, default=0, help="genome length (default: 3 x reads, i.e. 50x coverage at 150 bp: SURVEY 8d)")
    ap.add_argument("--seed", type=int, default=0, help="generator seed (default: 3 for the 50 M configuration, 2 otherwise: SURVEY 8d)")
    ap.add_argument("--err-ppm", type=int, default=0)
    ap.add_argument("--no-c2", action="store_true", help="skip the secondary line: BASELINE.json configs[1] (10 M reads)")
    ap.add_argument("--cpu-sample-reads", type=int, default=2_000_000)
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--cpu-port", action="store_true", help="time the CPU restatement instead of oracle/_ref")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-full", action="store_true", help="cpu_baseline on the WHOLE of BASELINE.json configs[1] (10 M reads) instead of the bounded sample: several minutes of host time, one-thread leg skipped")
    ap.add_argument("--no-scaling-model", action="store_true", help="skip the multi-GPU model and the extra build + probe pass without minimiser groups it measures (profiling runs)")
    ap.add_argument("--no-noisy-variant", action="store_true", help="skip the secondary line: the same workload with 0.1 %% substitution errors (SURVEY 8d)")
    ap.add_argument("--cpu-baseline-worker", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if not args.genome:
        args.genome = 3 * args.reads
    if not args.seed:
        args.seed = 3 if args.reads == 50_000_000 else 2

    if args.cpu_baseline_worker:
        import fixtures as fx
        import sage2_amd as s2
        print(json.dumps(cpu_baseline(args, s2, fx)), flush=True)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: this process becomes the launcher (it never touches HIP or torch.cuda) and starts
        # one child per GPU with the torchrun environment; rank 0's JSON line is relayed, every child's stderr passes through
        sys.exit(spawn_ranks(args.gpus))

    import numpy as np
    import torch
    import fixtures as fx
    import sage2_amd as s2

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    # SAGE2OV_BENCH_FORCE_SHARDED=1: one rank, but through the multi-rank code path (process group, record / flag / edge collectives on RCCL):
    # what `torchrun --nproc-per-node 1` rehearses on a one-GPU box before the first real multi-GPU run
    sharded = world > 1 or os.environ.get("SAGE2OV_BENCH_FORCE_SHARDED", "0") not in ("", "0")
    if sharded:
        import torch.distributed as dist
        for kk, vv in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29577"), ("RANK", "0"), ("WORLD_SIZE", "1")):
            os.environ.setdefault(kk, vv)                           # (forced one-rank run without a launcher)
        backend = os.environ.get("SAGE2OV_BENCH_BACKEND", "nccl")      # "gloo": rehearsal of the N>1 path on a box with fewer GPUs than ranks
        if backend != "nccl":
            local = local % torch.cuda.device_count()
            dist.init_process_group(backend)
        else:
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    assert torch.cuda.is_available(), "bench.py needs a GPU: the hot path has no CPU fallback"
    assert args.gpus == world, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}, or without a launcher (bench.py then starts its own ranks)"
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    pd = dict(seed=args.seed, genome_len=args.genome, n_reads=args.reads, read_len=args.read_len, err_ppm=args.err_ppm)
    p = fx.synth_params(pd)
    t0 = time.time()
    ctx = s2.Context(args.k, device=local, rank=rank, world=world)
    if rank == 0:
        g = s2.synth_genome(p)
        ctx.reads_add_synth(p, g)
        ctx.reads_organize()
        st = ctx.reads_stats()
        log(f"[bench] rank 0: {st.good_reads} good reads -> {st.unique_reads} unique, generated + organised in {time.time() - t0:.1f} s "
            f"(step 1 on the device: {ctx.timings().organize_ms:.1f} ms)")
    if sharded:
        # rank 0 organised the reads; everybody else imports the HBM image (broadcast over RCCL)
        meta = torch.zeros(6, dtype=torch.int64, device=dev)
        if rank == 0:
            meta = torch.tensor([st.unique_reads, st.words_per_read, st.max_read_length, st.good_reads, st.total_bp, 0], dtype=torch.int64, device=dev)
        dist.broadcast(meta, 0)
        n_u, wpr, mlen, good, bp, _ = meta.cpu().tolist()
        if rank == 0:
            words, freq = ctx.reads_export_words()
            tw = torch.from_numpy(words.view(np.int64)).to(dev)
        else:
            tw = torch.empty((n_u + 1) * wpr, dtype=torch.int64, device=dev)
        if dist.get_backend() == "gloo":
            th = tw.cpu(); dist.broadcast(th, 0); tw = th
        else:
            dist.broadcast(tw, 0)
        if rank != 0:
            ctx.reads_import_words(tw.cpu().numpy().view(np.uint64), n_u, wpr, mlen, np.ones(n_u + 1, dtype=np.uint16), good, bp)
        del tw
    st = ctx.reads_stats()

    def step():
        if sharded:
            from sage2_amd.dist import run_steps23_sharded
            run_steps23_sharded(ctx, dev)
        else:
            ctx.run_steps23()

    def fence():
        if sharded:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t1 = time.perf_counter()
    phase = dict(index_ms=0.0, probe_ms=0.0, reciprocal_ms=0.0, reduce_ms=0.0, convert_ms=0.0, reciprocal_cond_ms=0.0, reduce_marks_ms=0.0)
    pk_ms, pk_n, pk_f = 0.0, 0, 0
    for _ in range(args.steps):
        step()
        tm = ctx.timings()
        for kph in phase:
            phase[kph] += getattr(tm, kph)
        pk_ms += tm.probe_kernel_ms
        pk_n += tm.probe_kernel_launches
        pk_f += tm.probe_fast_launches
    fence()
    elapsed = time.perf_counter() - t1
    if sharded:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ost = ctx.overlap_stats()
    import digests as dg
    ed = ctx.edges()
    edges_crc = dg.edges_digest(ed["from"], ed["to"], ed["type"], ed["length"], ed["length_twin"])["edges_crc32"]   # canonical edge list: same at every N
    del ed
    # the oracle's numbers for this exact input (tests/golden/*_digest.json, oracle/make_digests.py): N, N_ov, edge count and edge-list crc32
    dname, want = dg.lookup(args.k, pd)
    digest_checked, bad = None, []
    if want is not None:
        got = dict(n_unique=st.unique_reads, n_ov=ost.verified_overlaps, edges=ost.edges, edges_crc32=edges_crc)
        bad = dg.compare(got, want, keys=list(got))
        digest_checked = dname
    if sharded:
        # every rank takes part in the collectives BEFORE anybody asserts: a mismatch on one rank must fail the run, not hang the others
        t = torch.tensor([edges_crc, -edges_crc, 0 if bad else 1], dtype=torch.int64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        cmin, cmax, all_ok = int(t[0].item()), -int(t[1].item()), int(t[2].item())
        assert cmin == cmax == edges_crc, "ranks disagree on the final edge list"
        assert all_ok == 1 or bad, f"another rank's result differs from the oracle digest {dname}"
    assert not bad, f"result differs from the oracle digest {dname}: " + "; ".join(bad)
    ms_per_step = 1e3 * elapsed / args.steps
    value = ost.verified_overlaps / (elapsed / args.steps)

    cfg_name = {10_000_000: "BASELINE.json configs[1]", 50_000_000: "BASELINE.json configs[2]/[3]"}.get(args.reads, "scaled-down BASELINE.json configs[1]") \
        if (args.read_len, args.k) == (150, 40) else "custom"
    if rank == 0:
        a_total, a_probe = algorithmic_bytes(st.unique_reads, ost.verified_overlaps, ost.edges, args.read_len, args.k)
        lo, hi = ctx.shard_range()
        share = (hi - lo) / max(st.unique_reads, 1)                  # this rank's share of the probe work
        kern_ms = pk_ms / max(pk_n, 1)
        ach = a_probe * share / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
        # memory-side bytes of the dominant kernel are NOT measured in this run: they are the PMC figure recorded under profiles/ for this
        # workload (error-free, one GPU) -- reported with its source, null for anything else
        traffic, traffic_src = None, None
        tj = os.path.join(ROOT, "profiles", "probe_traffic.json")
        if os.path.exists(tj) and args.err_ppm == 0 and world == 1:
            try:
                te = json.load(open(tj)).get(f"{args.reads}x{args.read_len}_k{args.k}_seed{args.seed}", {})
                sha = csrc_sha1()
                if te.get("kernel_source_sha1") == sha:              # a PMC pass of THIS device source (every kernel file and the launchers: grids and access paths move the traffic too); anything older is stale: null
                    traffic, traffic_src = te.get("bytes_per_launch"), te.get("source")
            except Exception:
                traffic = None
        # the bound the kernel is actually on: vector-instruction issue.  Wave-instructions per read come from a committed PMC pass of THIS kernel source
        # (profiles/probe_insts.json, keyed by a hash of the kernel's source: stale -> null); peak = one VALU wave-instruction per SIMD every 2 cycles
        # (MI355X_MICROARCH.md: SIMD-32, a wave64 instruction issues over 2 cycles) x 4 SIMDs x 256 CUs x 2.4 GHz
        valu = None
        ij = os.path.join(ROOT, "profiles", "probe_insts.json")
        if os.path.exists(ij) and args.err_ppm == 0 and world == 1 and kern_ms > 0:
            try:
                ie = json.load(open(ij)).get(f"{args.reads}x{args.read_len}_k{args.k}_seed{args.seed}", {})
                sha = csrc_sha1()
                if ie.get("kernel_source_sha1") == sha:
                    peak = 256 * 4 * 2.4e9 / 2.0
                    achieved = ie["valu_per_read"] * st.unique_reads / (kern_ms * 1e-3)
                    valu = {"wave_insts_per_read": ie["valu_per_read"], "scalar_insts_per_read": ie["salu_per_read"], "lds_insts_per_read": ie["lds_per_read"], "vmem_insts_per_read": ie["vmem_rd_per_read"],
                            "issue_peak": peak, "achieved": achieved, "unit": "VALU wave-instructions/s", "frac": achieved / peak,
                            "note": "peak counts every VALU instruction at 2 cycles; three-source, 64-bit and quarter-rate integer instructions cost more (DESIGN 5.2)", "source": ie.get("source")}
            except Exception:
                valu = None
        res = {
            "metric": "overlaps/sec + edge-set bit-identity vs OpenMP ref, 150 bp reads k=40",
            "value": value, "unit": "overlaps/s", "n_gpus": world, "world_size": (dist.get_world_size() if dist is not None else 1),
            "collectives": (dist.get_backend() if dist is not None else None), "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": f"{args.reads} x {args.read_len} bp synthetic paired-end reads, k={args.k}, {args.genome} bp uniform random genome, "
                                   f"seed {args.seed}, err {args.err_ppm} ppm ({cfg_name})",
                       "unique_reads": st.unique_reads, "verified_overlaps": ost.verified_overlaps, "edges": ost.edges, "edges_crc32": edges_crc,
                       "unresolved_reads": ost.left_to_explore, "partition": f"locality-order position range x{world}" if sharded else "single GPU",
                       "oracle_digest_asserted": digest_checked,
                       # the digests are generated by the CPU restatement (oracle/make_digests.py); where oracle/pin_reference.py has run the REFERENCE BINARY on
                       # the same reads, its P.graph3 (md5, size) and log counters are in the digest too and equal the restatement's
                       "oracle_digest_generated_by": (want or {}).get("generated_by"),
                       "oracle_digest_pinned_by_reference_binary": bool((want or {}).get("reference_binary_graph3_identical")),
                       "timed_region": "index build + initial pass + reduce + sort/convert; reads resident in HBM"},
            "phases_ms": {kph: v / args.steps for kph, v in phase.items()},
            "reads_per_s": st.unique_reads / (elapsed / args.steps),
            "roofline": {"bound": "hbm", "kernel": "k_probe_fast", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src and f"recorded: {traffic_src} (rocprofv3 PMC pass of this workload, not of this run)",
                         # memory-side bytes per launch (PMC, profiles/probe_traffic.json) over the live kernel time: what the kernel really pulls
                         "traffic_achieved": (traffic * share / (kern_ms * 1e-3) / 1e9) if (traffic and kern_ms > 0) else None,
                         "traffic_frac": (traffic * share / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if (traffic and kern_ms > 0) else None,
                         "limiter": "instruction issue, with the latency of dependent memory round trips behind it: since round 4 the kernel answers 93 % of the reads in RUN MODE (off a verified frame and a ring of verified slots, up to 16 reads per step, strand changes by mirroring the ring: DESIGN.md 5.2) -- 258 VALU + 214 scalar wave-instructions and 19.9 memory-side line requests of 128 B per read (round 3: 691 + 484 and 62), i.e. the memory side moves x0.62 of the algorithmic bytes at 4.0 TB/s; the PMC puts the VALU at 74 % and the scalar unit at 61 % busy (tools/pmc_wait.sh); 39 % of the wave cycles are the steps' three dependent round trips (table pair -> csr -> candidate dwords), 31 % the one read per minimiser (6.6 % of the reads) that takes the general path (111 table look-ups, 62 gathered candidates)",
                         # a probe pass is up to three launches of the kernel: a sample of 1/128 of the range, the rest (the instantiation the sample picked), and the
                         # few reads the first two listed; kernel_ms and the bytes are those of the whole pass (sum over its launches)
                         "valu": valu,
                         "kernel_ms": kern_ms, "kernel_launches_per_pass": pk_f / max(pk_n, 1), "algorithmic_bytes_per_launch": a_probe * share,
                         "whole_path_achieved": a_total / (elapsed / args.steps) / 1e9, "whole_path_frac": a_total / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS},
        }
        # the second-largest phase gets a roofline block of its own: the index build (locality order + store, tuples, partition, windows) against its
        # algorithmic bytes N (B + 4 x 16) -- read every read once, write four slots -- and, when a PMC pass of THIS source is on record, its memory-side bytes
        B_ = (args.read_len + 3) // 4
        a_index = st.unique_reads * (B_ + 4 * SLOT_ALG)
        ix_ms = res["phases_ms"]["index_ms"]
        ix_traffic, ix_src = None, None
        xj = os.path.join(ROOT, "profiles", "index_traffic.json")
        if os.path.exists(xj) and args.err_ppm == 0 and world == 1:
            try:
                xe = json.load(open(xj)).get(f"{args.reads}x{args.read_len}_k{args.k}_seed{args.seed}", {})
                if xe.get("kernel_source_sha1") == csrc_sha1():
                    ix_traffic, ix_src = xe.get("bytes_per_step"), xe.get("source")
            except Exception:
                ix_traffic = None
        res["roofline_index"] = {"bound": "hbm", "kernel": "index build: k_minimizer, k_pt_hist / k_pt_scatter (order: 5 passes, table: 2), k_loc_index, k_loc_scatter, k_ix_tuples, k_ix_window",
                                 "ms": ix_ms, "algorithmic_bytes_per_step": a_index, "achieved": (a_index / (ix_ms * 1e-3) / 1e9) if ix_ms > 0 else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": (a_index / (ix_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if ix_ms > 0 else None, "traffic": ix_traffic,
                                 "traffic_source": ix_src and f"recorded: {ix_src} (rocprofv3 PMC passes of this workload, not of this run)",
                                 "traffic_frac": (ix_traffic / (ix_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if (ix_traffic and ix_ms > 0) else None,
                                 "limiter": "a sort: the passes are bound per tuple (stable ranking: nine ballots per 64 tuples, two dependent LDS round trips), not per byte (DESIGN.md 5.1)"}
        def step4_of(c):
            """step 4 on the graph the timed steps left in HBM: outside the timed region, two runs, the second reported"""
            t4 = time.perf_counter(); c.graph_simplify(); w41 = time.perf_counter() - t4
            t4 = time.perf_counter(); c.graph_simplify(); w4 = time.perf_counter() - t4; s4 = c.simplify_stats()     # (the second call first frees the first one's result)
            return {"device_ms": s4.device_ms, "wall_ms_first_call": 1e3 * w41, "wall_ms": 1e3 * w4, "nodes_contracted": s4.nodes_contracted, "removed": s4.removed, "loop_iterations": s4.loop_iterations,
                    "edges_left": s4.edges, "reads_on_edges": s4.reads_on_edges}
        if world == 1 and not sharded and not args.no_scaling_model:
            # what a rank of a multi-GPU run would do differently: nothing at this coverage; if this run forces the minimiser groups on (SAGE2OV_MINIMIZER_INDEX=1), the pass without them is measured here (one untimed build + probe pass with the groups switched off), then the model
            alt = None
            if os.environ.get("SAGE2OV_MINIMIZER_INDEX") == "1" and args.err_ppm == 0:
                saved = os.environ.get("SAGE2OV_MINIMIZER_INDEX")
                os.environ["SAGE2OV_MINIMIZER_INDEX"] = "0"
                try:
                    ctx.options_reload()                             # (the library reads its switches once, at context creation)
                    ctx.timings_reset(); ctx.index_build(); ctx.overlap_probe_shard(); ta = ctx.timings()
                    alt = {"index_ms": ta.index_ms, "probe_ms": ta.probe_ms}
                finally:
                    if saved is None:
                        os.environ.pop("SAGE2OV_MINIMIZER_INDEX", None)
                    else:
                        os.environ["SAGE2OV_MINIMIZER_INDEX"] = saved
                    ctx.options_reload()
                ctx.run_steps23()                                    # (back to the state the step-4 figures below start from)
            res["scaling_model"] = scaling_model(st.unique_reads, ost.edges, ms_per_step, res["phases_ms"], alt)
            if alt:
                res["scaling_model"]["without_minimiser_groups_ms"] = alt
        if world == 1 and not args.no_step4:
            res["step4"] = step4_of(ctx)
        def side_workload(spd, nsteps, with_step4=False):
            """a secondary workload on its own context, outside the headline's timed region: one warm-up step, `nsteps` timed steps (same
            timed region as the headline), its result asserted against the oracle digest when one is committed for these parameters"""
            sp = fx.synth_params(spd)
            c = s2.Context(args.k, device=local)
            c.reads_add_synth(sp, s2.synth_genome(sp)); c.reads_organize()
            c.run_steps23(); torch.cuda.synchronize()
            t_ = time.perf_counter(); ph = dict(index_ms=0.0, probe_ms=0.0, reciprocal_ms=0.0, reduce_ms=0.0, convert_ms=0.0, reciprocal_cond_ms=0.0, reduce_marks_ms=0.0); pkm, pkn = 0.0, 0
            for _ in range(nsteps):
                c.run_steps23(); tm_ = c.timings()
                for kph in ph:
                    ph[kph] += getattr(tm_, kph)
                pkm += tm_.probe_kernel_ms; pkn += tm_.probe_kernel_launches
            torch.cuda.synchronize(); per = (time.perf_counter() - t_) / nsteps
            o_, st_ = c.overlap_stats(), c.reads_stats()
            e_ = c.edges(); crc_ = dg.edges_digest(e_["from"], e_["to"], e_["type"], e_["length"], e_["length_twin"])["edges_crc32"]; del e_
            dn_, dw_ = dg.lookup(args.k, spd)
            if dw_ is not None:
                g_ = dict(n_unique=st_.unique_reads, n_ov=o_.verified_overlaps, edges=o_.edges, edges_crc32=crc_, edges_inserted=o_.edges_inserted, transitive_removed=o_.transitive_removed)
                bad_ = dg.compare(g_, dw_, keys=list(g_))
                assert not bad_, f"result differs from the oracle digest {dn_}: " + "; ".join(bad_)
            a_tot, a_pr = algorithmic_bytes(st_.unique_reads, o_.verified_overlaps, o_.edges, spd["read_len"], args.k)
            out = {"workload": f"{spd['n_reads']} x {spd['read_len']} bp, k={args.k}, {spd['genome_len']} bp genome, seed {spd['seed']}, err {spd.get('err_ppm', 0)} ppm"
                               + (f", repeats {spd['n_repeat_families']} x {spd['repeat_copies']} x {spd['repeat_len']} bp" if spd.get("n_repeat_families") else ""),
                   "ms_per_step": 1e3 * per, "value": o_.verified_overlaps / per, "unit": "overlaps/s", "steps": nsteps, "unique_reads": st_.unique_reads,
                   "verified_overlaps": o_.verified_overlaps, "edges": o_.edges, "edges_crc32": crc_, "unresolved_reads": o_.left_to_explore, "long_buckets": c.index_stats().long_buckets,
                   "oracle_digest_asserted": dn_, "phases_ms": {kph: v / nsteps for kph, v in ph.items()},
                   "probe_kernel_ms": pkm / max(pkn, 1), "probe_kernel_algorithmic_frac": (a_pr / (pkm / max(pkn, 1) * 1e-3) / 1e9 / HBM_PEAK_GBS) if pkm > 0 else None}
            out["scaling_model"] = scaling_model(st_.unique_reads, o_.edges, 1e3 * per, out["phases_ms"])
            if with_step4 and not args.no_step4:
                out["step4"] = step4_of(c)
            c.close()
            return out
        side = dict(seed=2, genome_len=30_000_000, n_reads=10_000_000, read_len=args.read_len) if args.reads == 50_000_000 else dict(pd)
        side.pop("err_ppm", None)
        if world == 1 and args.err_ppm == 0 and (not args.no_c2 or not args.no_noisy_variant):
            ctx.close()
        if world == 1 and not args.no_c2 and args.err_ppm == 0 and args.reads == 50_000_000:
            # BASELINE.json configs[1] (the configuration round 1 reported as the headline), three timed steps
            res["c2"] = side_workload(side, 3)
        if world == 1 and not args.no_noisy_variant and args.err_ppm == 0:
            # secondary workload of SURVEY 8(d): configs[1] with 0.1 % substitution errors (98 % of the reads then go through the reduce phase)
            res["noisy_variant"] = side_workload(dict(side, err_ppm=1000), 2, with_step4=True)
            # third workload: read errors AND high-copy repeats (k-mers with >= 100 copies are hidden by the index, hashTable.cpp:111-123, which
            # makes discovery one-sided: the reduce phase then needs the exploration order); one timed step
            res["repeat_variant"] = side_workload(dict(side, err_ppm=1000, n_repeat_families=10, repeat_copies=300, repeat_len=400), 1)
        if world == 1 and not args.no_cpu_baseline:
            ctx.close()
            res["cpu_baseline"] = cpu_baseline_subprocess(args)
            if res["cpu_baseline"]:
                res["speedup_vs_cpu_baseline"] = value / res["cpu_baseline"]["value"]
                rb = (want or {}).get("reference_binary")
                if rb:
                    # the reference binary on THIS input (not a sample): recorded when the digest was pinned (oracle/pin_reference.py, build container) -- the
                    # "same input" CPU figure of SURVEY 8(d), from the reference's own log (1-s resolution); not measured on this host, not used for the speed-up
                    fs = rb.get("function_seconds", {})
                    t23 = sum(fs.get(kk, 0) for kk in ("hashPrefixesAndSuffix", "buildInitialOverlapGraph", "buildOverlapGraphEconomy", "sortEdgesEconomy", "convertGraph"))
                    if t23:
                        res["cpu_baseline"]["reference_on_this_input"] = {"value": ost.verified_overlaps / t23, "unit": "overlaps/s", "seconds_steps_2_3": t23, "function_seconds": fs,
                                                                         "threads": rb.get("threads"), "peak_rss_gb": rb.get("peak_rss_gb"), "where": "recorded: build container, oracle/pin_reference.py (not this host)"}
        print(json.dumps(res), flush=True)
    if sharded:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
