"""GPU: the multi-rank orchestration (sage2_amd/dist.py::run_steps23_sharded) itself, not only the C-ABI exchange points: `world`
processes share the one GPU of the test box, collectives staged through gloo.  Mixed read lengths (containment marks that one rank's
probe places on reads of another rank, economyGraph.cpp:735), noisy data (reduce phase) and long buckets (ranked reduce)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("name,world", [("g5_mixedlen_k21", 2), ("g5_mixedlen_k21", 3), ("g3_noisy_rep_k21", 2), ("g4_highcopy_k21", 3)])
def test_run_steps23_sharded_matches_reference(name, world, tmp_path):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(29531 + world), os.path.join(ROOT, "tests", "dist_gpu_worker.py"), name, str(tmp_path)]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env, timeout=600)
    out = r.stdout.decode()
    assert r.returncode == 0 and f"DIST_GPU_OK {world}" in out, out[-3000:]
