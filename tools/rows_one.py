"""Profiling aid: 20 launches of one block-norm / per-channel rows case.  argv: dtype(f32|bf16|f16) block_size mode"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fp8-emulation-toolkit_amd"))
from mpemu.pytquant.hip import _binding as hip  # noqa: E402

hip._SO = os.environ.get("FPEMU_LIB") or hip._SO
dt = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}[sys.argv[1]]
bs = int(sys.argv[2])
mode = sys.argv[3] if len(sys.argv) > 3 else "E4M3_RNE"
n = 1 << 26
x = (torch.randn(n, device="cuda") * 3).to(dt)
y = torch.empty_like(x)
f, r = hip.parse_mode(mode)
reps = int(os.environ.get("REPS", "20"))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    hip.quantize_blocks_(x, y, bs, f, r)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print("%s block %d %s: events over %d back-to-back launches %.1f us/launch -> %.0f GB/s" % (sys.argv[1], bs, mode, reps, ms * 1e3, 2 * x.element_size() * n / ms / 1e6), flush=True)
