"""Multi-GPU orchestration of steps 2-3: one process per GPU, torch.distributed (backend "nccl" = RCCL over
xGMI on the GPU box, "gloo" in CPU tests).  The read set and the index are replicated; read ids are
range-partitioned for the probe/verify kernel.  Exchange steps -- all bulk, one-shot:

  1. all-gather of the fixed-size per-read records (right ext, left ext, connections, flags: 16 B/read):
     the reciprocal test reads the NEIGHBOUR's record (economyGraph.cpp:460);
  2. MAX all-reduce (= OR) of the two containment bit planes: a containment mark (economyGraph.cpp:735)
     lands on a read of any rank;
  3. all-gather of the variable-size per-rank edge buckets (counts first, then max-padded buffers);
  4. after the reduce phase (its marks are sharded over the ranks): all-gather of the per-rank survivor buckets, SUM of the removal counters.

The tensor plumbing below is backend-agnostic (it is what the gloo tests exercise); `run_steps23_sharded`
binds it to a sage2_amd.Context.
"""
import torch
import torch.distributed as dist

from .shard import RECORD_BYTES, EDGE_BYTES, shard_range, max_shard, pack_records, unpack_records  # noqa: F401  (re-exported)


def _sync(t: torch.Tensor):
    """collectives (and torch's own fills/copies) run on torch's stream, the library on its own:
    fence before handing buffers over, in either direction"""
    if t.is_cuda:
        torch.cuda.current_stream(t.device).synchronize()


def _staged(group=None):
    """gloo has no all-gather for device tensors: a rehearsal of the multi-rank path on a box without RCCL peers
    (e.g. 2 ranks sharing one GPU) stages the collectives through the host.  Never taken with "nccl"."""
    return dist.get_backend(group) == "gloo"


def _all_gather(recv, send, group=None):
    if send.is_cuda and _staged(group):
        r = torch.empty(recv.shape, dtype=recv.dtype)
        dist.all_gather_into_tensor(r, send.cpu(), group=group)
        recv.copy_(r)
    else:
        dist.all_gather_into_tensor(recv, send, group=group)


def allgather_records(send: torch.Tensor, n_unique: int, group=None):
    """send: uint8 [max_shard*RECORD_BYTES] holding this rank's records (padded).  Returns a list of (first_id, n, tensor)."""
    world = dist.get_world_size(group)
    recv = torch.empty(world * send.numel(), dtype=torch.uint8, device=send.device)
    _all_gather(recv, send, group)
    _sync(recv)
    out = []
    for r in range(world):
        lo, hi = shard_range(n_unique, r, world)
        out.append((lo, hi - lo, recv[r * send.numel(): r * send.numel() + (hi - lo) * RECORD_BYTES]))
    return out


def allreduce_flags(planes: torch.Tensor, group=None):
    if planes.is_cuda and _staged(group):
        h = planes.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.MAX, group=group)
        planes.copy_(h)
    else:
        dist.all_reduce(planes, op=dist.ReduceOp.MAX, group=group)
    _sync(planes)
    return planes


def allgather_edge_buckets(bucket: torch.Tensor, n_edges: int, group=None, cap_edges: int = None):
    """bucket: uint8 tensor with this rank's n_edges*16 bytes (may be longer).  Returns the concatenation of all ranks' buckets in
    rank order and the total edge count.  ONE collective: every rank sends a fixed-size buffer = a 16-byte header (its edge count) +
    `cap_edges` edge slots, so no count exchange (and no device-to-host round trip) precedes the payload.  cap_edges must bound every
    rank's bucket; the caller knows such a bound (a rank emits at most 2 edges per read it owns in the reciprocal pass).  Without a
    bound the counts are exchanged first (two collectives)."""
    world = dist.get_world_size(group)
    if cap_edges is None:
        cnt = torch.tensor([n_edges], dtype=torch.int64, device=bucket.device)
        cnts = torch.empty(world, dtype=torch.int64, device=bucket.device)
        _all_gather(cnts, cnt, group)
        cap_edges = max(int(cnts.max().item()), 1)
    assert n_edges <= cap_edges, "edge bucket larger than the agreed capacity"
    slot = (1 + cap_edges) * EDGE_BYTES
    send = torch.zeros(slot, dtype=torch.uint8, device=bucket.device)
    send[:8] = torch.tensor([n_edges], dtype=torch.int64).view(torch.uint8).to(bucket.device)
    send[EDGE_BYTES: EDGE_BYTES + n_edges * EDGE_BYTES] = bucket[: n_edges * EDGE_BYTES]
    recv = torch.empty(world * slot, dtype=torch.uint8, device=bucket.device)
    _all_gather(recv, send, group)
    _sync(recv)
    heads = recv.view(world, slot)[:, :8].contiguous().cpu().view(torch.int64).reshape(world).tolist()      # (after the payload has arrived: no extra collective)
    parts = [recv[r * slot + EDGE_BYTES: r * slot + EDGE_BYTES + heads[r] * EDGE_BYTES] for r in range(world)]
    return torch.cat(parts) if parts else recv[:0], sum(heads)


def run_steps23_sharded(ctx, device, group=None):
    """The timed region (index build + initial + reduce + convert) on `world` GPUs.  Every rank ends up with the
    complete canonical edge list."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n = ctx.reads_stats().unique_reads
    ctx.timings_reset()
    ctx.index_build()
    ctx.overlap_probe_shard()
    ms = max_shard(n, world)
    send = torch.zeros(ms * RECORD_BYTES, dtype=torch.uint8, device=device)
    _sync(send)                                   # the fill runs on torch's stream, the export on the library's
    ctx.shard_export_records(send.data_ptr(), ms)
    # the containment marks this rank's probe placed on reads of ANY rank leave before other ranks' records come in
    # (the import ORs the flags it carries, so the order is not load-bearing; it mirrors the C-ABI test)
    planes = torch.zeros(ctx.shard_flags_bytes(), dtype=torch.uint8, device=device)
    _sync(planes)
    ctx.shard_export_flags(planes.data_ptr())
    for first, cnt, t in allgather_records(send, n, group):
        if cnt:
            t = t.contiguous()
            ctx.shard_import_records(t.data_ptr(), first, cnt)
    allreduce_flags(planes, group)
    ctx.shard_import_flags(planes.data_ptr())
    ctx.overlap_reciprocal()
    ne = ctx.shard_edges_count()
    bucket = torch.zeros(max(ne, 1) * EDGE_BYTES, dtype=torch.uint8, device=device)
    _sync(bucket)
    ctx.shard_edges_export(bucket.data_ptr(), max(ne, 1))
    # (counts first, then buckets padded to the largest: the only a-priori bound -- two edges per read of the range -- would double the bytes
    # of the one-collective form, which costs more over xGMI than the 8-byte count exchange it saves)
    allb, total = allgather_edge_buckets(bucket, ne, group)
    allb = allb.contiguous()
    _sync(allb)
    ctx.shard_edges_set(allb.data_ptr() if total else 0, total)
    # reduce phase: hit lists and adjacency of all unresolved reads on every rank, the marks (economyGraph.cpp:643-707) for this rank's share only;
    # the surviving edges of the shares are exchanged like the edge buckets, the removal counters summed
    ctx.overlap_reduce()
    ns, rem = ctx.shard_survivors_count()
    sb = torch.zeros(max(ns, 1) * EDGE_BYTES, dtype=torch.uint8, device=device)
    _sync(sb)
    ctx.shard_survivors_export(sb.data_ptr(), max(ns, 1))
    alls, stotal = allgather_edge_buckets(sb, ns, group)
    remt = torch.tensor([rem], dtype=torch.int64, device=device)
    if remt.is_cuda and _staged(group):
        h = remt.cpu(); dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group); remt = h
    else:
        dist.all_reduce(remt, op=dist.ReduceOp.SUM, group=group)
    alls = alls.contiguous()
    _sync(alls)
    ctx.shard_survivors_set(alls.data_ptr() if stotal else 0, stotal, int(remt.item()))
    ctx.overlap_convert()
