"""Worker for tests/test_gpu_dist.py: sage2_amd.dist.run_steps23_sharded end to end with `world` ranks that share cuda:0 (collectives
staged through gloo: the code path of the multi-GPU bench minus RCCL), or -- backend "nccl", one rank -- with every collective on
RCCL and device tensors (RCCL refuses two ranks on one GPU, so that is the largest world a one-GPU box can run).  Every rank must end
with the reference's P.graph3."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fixtures as fx                       # noqa: E402
import sage2_amd as s2                      # noqa: E402
from sage2_amd.dist import run_steps23_sharded   # noqa: E402


def main():
    name, out = sys.argv[1], sys.argv[2]
    backend = sys.argv[3] if len(sys.argv) > 3 else "gloo"
    dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    m = fx.golden(name)
    bases, off = fx.make_reads(m["synth"])
    ctx = s2.Context(m["k"], device=0, rank=rank, world=world)
    ctx.reads_add_ascii(bases, off); ctx.reads_organize()
    for _ in range(2):                       # twice: the second pass starts from the state the first one left
        run_steps23_sharded(ctx, dev)
    gp = os.path.join(out, f"r{rank}.graph3"); ctx.graph_save(gp)
    same = open(gp, "rb").read() == fx.golden_graph3(name)
    st = ctx.overlap_stats()
    ok = same and st.contained_extension == m["counters"]["contained_extension"] and st.contained_size == m["counters"]["contained_size"]
    if "transitive_removed" in m["counters"]:
        ok = ok and st.transitive_removed == m["counters"]["transitive_removed"]          # (summed over the ranks' shares of the reduce phase)
    if not ok:
        print(f"rank {rank}: graph3 identical {same}; contained {st.contained_extension}/{m['counters']['contained_extension']} {st.contained_size}/{m['counters']['contained_size']}; "
              f"edges {st.edges}; removed {st.transitive_removed}/{m['counters'].get('transitive_removed')}", flush=True)
    t = torch.tensor([1 if ok else 0], device=dev if backend == "nccl" else "cpu"); dist.all_reduce(t, op=dist.ReduceOp.MIN)
    if rank == 0:
        print("DIST_GPU_OK" if int(t.item()) == 1 else "DIST_GPU_MISMATCH", world, dist.get_backend(), flush=True)
    ctx.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
