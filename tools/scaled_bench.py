"""Fused per-tensor quantise (amax -> scale -> cast) by tensor size: the single-word path (zero + atomic
amax + cast, fpemu_quantize_per_tensor) against the workspace path (per-block maxima + cast,
fpemu_quantize_per_tensor_ws)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fp8-emulation-toolkit_amd"))
from mpemu.pytquant.hip import _binding as hip  # noqa: E402

hip._SO = os.environ.get("FPEMU_LIB", hip._SO)  # tuning aid: a library built with other geometry constants

f, r = hip.parse_mode("E4M3_RNE")
for mib in (0.25, 3, 16, 32, 64, 128, 256, 512, 1024):
    n = int(mib * (1 << 20)) // 4
    x = torch.randn(n, device="cuda")
    y = torch.empty_like(x)
    res = []
    for ws in (torch.zeros(1, dtype=torch.int32, device="cuda"), torch.zeros(257, dtype=torch.int32, device="cuda")):
        for _ in range(3):
            hip.quantize_per_tensor_(x, y, f, r, 448.0, amax_ws=ws)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            hip.quantize_per_tensor_(x, y, f, r, 448.0, amax_ws=ws)
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 50)
    print("%7.2f MiB  single word %.4f ms | workspace %.4f ms  (%7.1f GB/s algorithmic, 12 B/elem)" % (
        mib, res[0], res[1], 12 * n / res[1] / 1e6), flush=True)
