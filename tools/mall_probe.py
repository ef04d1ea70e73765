"""Does the cast that follows an amax pass read a mid-size tensor from the Infinity Cache?  Run under
rocprofv3 --kernel-trace --stats and compare the cast kernel's duration with and without the preceding amax."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fp8-emulation-toolkit_amd"))
from mpemu.pytquant.hip import _binding as hip  # noqa: E402

mib = float(sys.argv[1]) if len(sys.argv) > 1 else 128
which = sys.argv[2] if len(sys.argv) > 2 else "both"
n = int(mib * (1 << 20)) // 4
x = torch.randn(n, device="cuda")
y = torch.empty_like(x)
f, r = hip.parse_mode("E4M3_RNE")
ws = torch.zeros(257, dtype=torch.int32, device="cuda")
for _ in range(100):
    if which in ("both", "plain"):
        hip.cast_(x, y, f, r, scale=1.0)
for _ in range(100):
    if which in ("both", "scaled"):
        hip.quantize_per_tensor_(x, y, f, r, 448.0, amax_ws=ws)
torch.cuda.synchronize()
