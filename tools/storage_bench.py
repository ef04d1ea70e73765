"""Cast throughput by storage dtype and scale (128 Mi elements)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fp8-emulation-toolkit_amd"))
from mpemu.pytquant.hip import _binding as hip  # noqa: E402

hip._SO = os.environ.get("FPEMU_LIB", hip._SO)  # tuning aid: a library built with other geometry constants

n = 1 << 27
for dt, bpe in ((torch.float32, 8), (torch.float16, 4), (torch.bfloat16, 4)):
    x = (torch.randn(n, device="cuda") * 3).to(dt)
    y = torch.empty_like(x)
    for mode in ("E4M3_RNE", "E5M2_STOCHASTIC"):
        f, r = hip.parse_mode(mode)
        for scale in (1.0, 3.0):
            for _ in range(3):
                hip.cast_(x, y, f, r, scale=scale, seed=1, offset=2)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                hip.cast_(x, y, f, r, scale=scale, seed=1, offset=2)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 20
            print("%-9s %-16s scale %.0f  %.4f ms  %7.1f GB/s" % (str(dt).split(".")[1], mode, scale, ms,
                                                                  bpe * n / ms / 1e6), flush=True)
