"""Fused per-tensor quantise (amax + cast, fpemu_quantize_per_tensor) by tensor size: does the cast's
re-read come from the Infinity Cache?"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fp8-emulation-toolkit_amd"))
from mpemu.pytquant.hip import _binding as hip  # noqa: E402

f, r = hip.parse_mode("E4M3_RNE")
for mib in (4, 16, 32, 64, 96, 128, 160, 256, 1024):
    n = mib * (1 << 20) // 4
    x = torch.randn(n, device="cuda")
    y = torch.empty_like(x)
    ws = torch.zeros(1, dtype=torch.int32, device="cuda")
    for _ in range(3):
        hip.quantize_per_tensor_(x, y, f, r, 448.0, amax_ws=ws)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        hip.quantize_per_tensor_(x, y, f, r, 448.0, amax_ws=ws)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print("%5d MiB  %.4f ms  algorithmic (12 B/elem) %7.1f GB/s" % (mib, ms, 12 * n / ms / 1e6), flush=True)
