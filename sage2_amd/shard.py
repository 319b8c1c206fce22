"""Read-id range partition of the probe work -- plain Python, no torch: usable from processes that
only talk to the C ABI.  `sage2_amd.dist` binds these to torch.distributed."""

RECORD_BYTES = 24     # per-read record exchanged between ranks: right ext u64, left ext u64, connections u32, containment flags u32
EDGE_BYTES = 16       # edge candidate: from u32, to u32, length u32, type u32


def shard_range(n_unique, rank, world):
    """ids [lo, hi) of rank `rank`; identical to sage2ov_shard_range."""
    return 1 + (n_unique * rank) // world, 1 + (n_unique * (rank + 1)) // world


def max_shard(n_unique, world):
    return max(shard_range(n_unique, r, world)[1] - shard_range(n_unique, r, world)[0] for r in range(world))
