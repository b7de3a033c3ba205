"""GPU: the N > 1 path end to end on one card (two ranks, gloo): shard pack -> fused all-gather -> sharded ld_triangle."""
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.gpu
def test_two_ranks_one_card_sharded_triangle():
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), str(ROOT / "tests" / "_gpu_dist_worker.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=500, env=env, cwd=str(ROOT))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "GPU_DIST_OK" in r.stdout
