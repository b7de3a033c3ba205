"""CPU: the multi-rank path (slab shards -> all-gather -> unit ranges) with gloo, world_size 2."""
import os
import socket
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_rank_gloo_rehearsal():
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), str(ROOT / "tests" / "_gloo_worker.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=str(ROOT))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "GLOO_OK" in r.stdout


def test_bench_launcher_fails_loudly_without_gpu():
    """`python bench.py --gpus 2` starts the ranks itself (a child torch.distributed.run, before any GPU call).  On a box
    without a GPU the ranks refuse to run -- there is no CPU path -- and the launcher must hand that on as a non-zero exit
    code and no JSON line, after its one retry without the HIP graph."""
    import torch

    if torch.cuda.is_available():
        import pytest
        pytest.skip("a GPU is present: covered by tests/test_gpu_dist.py")
    env = dict(os.environ, OMP_NUM_THREADS="1")
    env.pop("RANK", None)
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--snps", "300",
                        "--backend", "gloo", "--deadline", "300"], capture_output=True, text=True, timeout=700, env=env,
                       cwd=str(ROOT))
    assert r.returncode != 0
    assert '"metric"' not in r.stdout
    assert "needs a HIP device" in r.stderr and "once more in the plainest mode" in r.stderr
