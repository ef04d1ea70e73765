"""Per-launch duration of the flat cast from a cold start (does the first few ms run at lower clocks?)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fp8-emulation-toolkit_amd"))
from mpemu.pytquant.hip import _binding as hip  # noqa: E402

hip.lib()
n = 1 << 28
x = torch.randn(n, device="cuda")
y = torch.empty_like(x)
f, r = hip.parse_mode("E4M3_RNE")
torch.cuda.synchronize()
K = 400
ev = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
ev[0].record()
for i in range(K):
    hip.cast_(x, y, f, r, scale=1.0)
    ev[i + 1].record()
torch.cuda.synchronize()
per = [ev[i].elapsed_time(ev[i + 1]) for i in range(K)]
for b in range(0, K, 20):
    blk = per[b:b + 20]
    print("launches %3d-%3d  avg %.4f ms  min %.4f  max %.4f  (%.0f GB/s)" % (b, b + 19, sum(blk) / 20, min(blk), max(blk),
                                                                           8 * n / (sum(blk) / 20) / 1e6), flush=True)
