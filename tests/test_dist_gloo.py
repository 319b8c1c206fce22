"""CPU: the multi-GPU exchange plumbing (sage2_amd/dist.py) on world_size 2 and 3 with the gloo backend."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world", [2, 3])
def test_exchange_plumbing_gloo(world):
    env = dict(os.environ, OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(29511 + world), os.path.join(ROOT, "tests", "dist_worker.py")]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env, timeout=600)
    out = r.stdout.decode()
    assert r.returncode == 0 and f"DIST_OK {world}" in out, out[-3000:]
