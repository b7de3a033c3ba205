#!/usr/bin/env python3
"""Peak-rate probes: AND+BCNT on the VALU and int8 MFMA on the matrix pipe, at several occupancies."""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from ld_tools_amd._lib import check, lib  # noqa: E402

dev = torch.device("cuda", 0)
cus = torch.cuda.get_device_properties(dev).multi_processor_count
s = torch.cuda.current_stream().cuda_stream


def timed(fn):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e-3


out = {}
iters = 20000
for variant, macs in ((0, 32768), (1, 16384)):
    for blocks_per_cu, threads in ((1, 256), (2, 256), (4, 256)):
        blocks = cus * blocks_per_cu
        sink = torch.empty(blocks * threads, dtype=torch.int32, device=dev)
        t = timed(lambda: check(lib.ldx_probe_mfma_dev(sink.data_ptr(), blocks, threads, iters, variant, s)))
        n_mfma = blocks * (threads // 64) * iters * 8
        waves_per_simd = blocks_per_cu * (threads // 64) / 4
        # cycles per MFMA per SIMD at an assumed 2.4 GHz (the real clock is lower under load)
        out[f"mfma_v{variant}_w{waves_per_simd:g}"] = {
            "TMAC_per_s": n_mfma * macs / t / 1e12,
            "ns_per_mfma_per_simd": t / (n_mfma / (cus * 4)) * 1e9}
for blocks_per_cu in (1, 2, 4):
    blocks, threads, it = cus * blocks_per_cu, 1024, 4000
    sink = torch.empty(blocks * threads, dtype=torch.int32, device=dev)
    t = timed(lambda: check(lib.ldx_probe_andpop_dev(sink.data_ptr(), blocks, threads, it, s)))
    out[f"andpop_w{blocks_per_cu * 4}"] = {"T_lane_ops_per_s": blocks * threads * it * 128 / t / 1e12}
print(json.dumps(out, indent=1))
