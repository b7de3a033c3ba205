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


def test_eight_rank_gloo_rehearsal():
    """The same worker with a process group of EIGHT (gloo, CPU): small panels whose slabs do not reach every rank (ranks
    with no rows at all), and configs[3]'s geometry -- 100 000 SNPs, 782 slabs as 98, ..., 98, 96 -- through the
    asynchronous fused exchange bench.py uses, with the unit ranges of the eight ranks checked against each other.
    (Eight ranks on ONE card is not something a GPU box of this pool allows -- at most six processes may have the card
    open, the test runner included -- so the GPU rehearsal of tests/test_gpu_dist.py runs four: the same uneven-shard
    code with a real kernel behind it.)"""
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), str(ROOT / "tests" / "_gloo_worker.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=str(ROOT))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "GLOO_OK" in r.stdout


def test_fused_exchange_places_the_shards_of_configs3():
    """BASELINE configs[3] geometry, 100 000 SNPs x 5008 haplotypes over 8 ranks: 782 slabs as 98, ..., 98, 96 -- the
    uneven case of fused_gather_finish (the rank-major concatenation is the full piece followed by padding).  The eight
    ranks' byte shards are built with fill_shard exactly as fused_gather_start fills them and laid side by side as the
    all-gather would; the placement must reproduce every slab image and every count, with and without the REF plane.
    (The collective itself runs in the two-rank rehearsal above and in tests/test_gpu_dist.py.)"""
    import numpy as np
    import torch

    sys.path.insert(0, str(ROOT))
    from ld_tools_amd import dist as ldist

    n_snps, n_hap, world = 100000, 5008, 8
    parts = ldist.slab_partition(n_snps, world)
    slabs = [(e - b + 127) // 128 for b, e in parts]
    assert slabs == [98] * 7 + [96] and sum(slabs) == 782
    slab_bytes = ldist.n_chunks(n_hap) * 128 * 16
    assert slab_bytes == 80 * 1024
    rng = np.random.default_rng(7)
    n_pad = sum(slabs) * 128
    for with_ref in (False, True):
        planes = {k: torch.from_numpy(rng.integers(0, 256, size=sum(slabs) * slab_bytes, dtype=np.uint8))
                  for k in (("alt", "ref") if with_ref else ("alt",))}
        acnt = torch.from_numpy(rng.integers(0, 5009, size=n_pad, dtype=np.int32))
        rcnt = torch.from_numpy(rng.integers(0, 5009, size=n_pad, dtype=np.int32))
        big, cnt_bytes, per_slab, *_ = ldist._gather_layout(slabs, slab_bytes, with_ref)
        recv = torch.full((world * big * per_slab,), 0xEE, dtype=torch.uint8)      # padding bytes must never be copied out
        off = 0
        for r, mine in enumerate(slabs):
            local = {k: v[off * slab_bytes: (off + mine) * slab_bytes] for k, v in planes.items()}
            local["acnt"] = acnt[off * 128: (off + mine) * 128]
            local["rcnt"] = rcnt[off * 128: (off + mine) * 128]
            ldist.fill_shard(recv[r * big * per_slab: (r + 1) * big * per_slab], local, mine, slabs, slab_bytes, with_ref)
            off += mine
        full = {k: torch.zeros_like(v) for k, v in planes.items()}
        full["acnt"] = torch.zeros(n_pad, dtype=torch.int32)
        full["rcnt"] = torch.zeros(n_pad, dtype=torch.int32)
        ldist.fused_gather_finish(full, (None, recv), slabs, slab_bytes)
        for k, v in planes.items():
            assert torch.equal(full[k], v), k
        assert torch.equal(full["acnt"], acnt) and torch.equal(full["rcnt"], rcnt)
    # and an arbitrary split (third branch) on a small geometry
    slabs = [2, 5, 0, 3]
    sb = 2 * 128 * 16
    alt = torch.arange(sum(slabs) * sb, dtype=torch.int64).to(torch.uint8)
    cnt = torch.arange(sum(slabs) * 128, dtype=torch.int32)
    big, cnt_bytes, per_slab, *_ = ldist._gather_layout(slabs, sb, False)
    recv = torch.full((len(slabs) * big * per_slab,), 0xEE, dtype=torch.uint8)
    off = 0
    for r, mine in enumerate(slabs):
        local = {"alt": alt[off * sb: (off + mine) * sb], "acnt": cnt[off * 128: (off + mine) * 128],
                 "rcnt": cnt[off * 128: (off + mine) * 128] + 1}
        ldist.fill_shard(recv[r * big * per_slab: (r + 1) * big * per_slab], local, mine, slabs, sb, False)
        off += mine
    full = {"alt": torch.zeros_like(alt), "acnt": torch.zeros_like(cnt), "rcnt": torch.zeros_like(cnt)}
    ldist.fused_gather_finish(full, (None, recv), slabs, sb)
    assert torch.equal(full["alt"], alt) and torch.equal(full["acnt"], cnt) and torch.equal(full["rcnt"], cnt + 1)


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


def test_bench_launcher_ends_the_whole_process_group_at_the_deadline():
    """on a --deadline overrun the launcher must end torch.distributed.run AND the rank grandchildren (they hold
    the GPUs), return 124 and start nothing beside them.  The ranks hang on request (--debug-hang, before any GPU call, so
    this runs on a box without a GPU); afterwards no process of the run is left."""
    import re
    import time

    import psutil

    env = dict(os.environ, OMP_NUM_THREADS="1")
    env.pop("RANK", None)
    env.pop("WORLD_SIZE", None)
    t0 = time.time()
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--snps", "300",
                        "--backend", "gloo", "--deadline", "25", "--debug-hang"], capture_output=True, text=True, timeout=300,
                       env=env, cwd=str(ROOT))
    assert r.returncode == 124, (r.returncode, r.stderr[-2000:])
    assert time.time() - t0 < 120
    assert "ending their process group" in r.stderr and "once more in the plainest mode" not in r.stderr
    pids = [int(x) for x in re.findall(r"pid (\d+) hangs on request", r.stderr)]
    assert len(pids) == 2, r.stderr[-2000:]
    time.sleep(1.0)
    for pid in pids:
        alive = psutil.pid_exists(pid) and psutil.Process(pid).status() != psutil.STATUS_ZOMBIE
        assert not alive, f"rank process {pid} survived the launcher"
