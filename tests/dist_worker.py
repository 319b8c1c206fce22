"""Worker for tests/test_dist_gloo.py: world_size-N gloo run of the exchange plumbing in sage2_amd/dist.py.
Each rank holds the oracle's full per-read results but only publishes its own id range; after the exchanges every
rank must hold the complete, identical picture (what the reciprocal pass and convert need)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fixtures as fx          # noqa: E402
import oracle_lib as ol        # noqa: E402
from sage2_amd import dist as sd   # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    m = fx.golden("g5_mixedlen_k21")          # has containment flags and unresolved reads
    bases, off = fx.make_reads(m["synth"])
    o = ol.Oracle(m["k"], 2)
    o.add_reads_ascii(bases, off); o.organize(); o.build_index(); o.initial()
    right, left, status, conn = o.export_initial()
    n = len(conn) - 1
    lo, hi = sd.shard_range(n, rank, world)
    # the union of all shard ranges is exactly 1..n
    ranges = [sd.shard_range(n, r, world) for r in range(world)]
    assert ranges[0][0] == 1 and ranges[-1][1] == n + 1 and all(ranges[r][1] == ranges[r + 1][0] for r in range(world - 1))
    # ---- 1. per-read records (RECORD_BYTES: right / left extension, connections, containment flags), own range only
    cflag = (np.arange(n + 1) * 7 + 1) % 4                  # (the oracle keeps containment in the status; any two bits must survive the wire)
    ms = sd.max_shard(n, world)
    send = torch.zeros(ms * sd.RECORD_BYTES, dtype=torch.uint8)
    mine = torch.from_numpy(sd.pack_records(right[lo:hi], left[lo:hi], conn[lo:hi], cflag[lo:hi]).copy())
    assert mine.numel() == (hi - lo) * sd.RECORD_BYTES
    send[: mine.numel()] = mine
    got = [np.zeros(n + 1, dtype=np.uint64), np.zeros(n + 1, dtype=np.uint64), np.zeros(n + 1, dtype=np.uint32), np.zeros(n + 1, dtype=np.uint32)]
    for first, cnt, t in sd.allgather_records(send, n):
        for a, b in zip(got, sd.unpack_records(t.numpy().tobytes())):
            a[first:first + cnt] = b
    assert np.array_equal(got[0][1:], np.asarray(right, dtype=np.uint64)[1:]) and np.array_equal(got[1][1:], np.asarray(left, dtype=np.uint64)[1:]), "record all-gather does not reassemble the extensions"
    assert np.array_equal(got[2][1:], np.asarray(conn, dtype=np.uint32)[1:]) and np.array_equal(got[3][1:], cflag[1:].astype(np.uint32)), "record all-gather loses connections or flags"
    # ---- 2. containment bit planes: every rank marks only what ITS reads contain -> OR over ranks
    is6 = (status == 6)
    planes = np.zeros(2 * (n + 1), dtype=np.uint8)
    idx = np.nonzero(is6)[0]
    sel = idx[idx % world == rank]                        # pretend rank r discovered every world-th containment
    planes[sel] = 1
    t = torch.from_numpy(planes)
    sd.allreduce_flags(t)
    assert np.array_equal(t.numpy()[: n + 1] != 0, is6), "flag all-reduce lost containment marks"
    # ---- 3. ragged edge buckets
    rng = np.random.default_rng(100 + rank)
    ne = int(rng.integers(0, 50)) if rank else 0          # rank 0 publishes an empty bucket
    bucket = rng.integers(0, 255, size=max(ne, 1) * sd.EDGE_BYTES, dtype=np.uint8)
    allb, total = sd.allgather_edge_buckets(torch.from_numpy(bucket), ne)
    cnts = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(cnts, torch.tensor([ne]))
    assert total == sum(int(c) for c in cnts) and allb.numel() == total * sd.EDGE_BYTES
    start = sum(int(c) for c in cnts[:rank]) * sd.EDGE_BYTES
    assert np.array_equal(allb.numpy()[start:start + ne * sd.EDGE_BYTES], bucket[: ne * sd.EDGE_BYTES])
    dist.barrier()
    if rank == 0:
        print("DIST_OK", world)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
