"""Tuning aid: 16-bit storage rows with short blocks (FPEMU_LIB selects a library build)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fp8-emulation-toolkit_amd"))
from mpemu.pytquant.hip import _binding as hip  # noqa: E402

hip._SO = os.environ.get("FPEMU_LIB") or hip._SO
n = 1 << 26


def timed(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for mode in os.environ.get("MODES", "E4M3_RNE E5M2_STOCHASTIC").split():
    out = []
    x = (torch.randn(n, device="cuda") * 3).to(torch.bfloat16)
    y = torch.empty_like(x)
    f, r = hip.parse_mode(mode)
    ms = timed(lambda: hip.cast_(x, y, f, r, scale=3.0, seed=1, offset=2))
    out.append("flat: %.0f" % (4 * n / ms / 1e6))
    for bs in (8, 16, 32, 64, 128, 256, 512, 1024, 4096):
        ms = timed(lambda: hip.quantize_blocks_(x, y, bs, f, r))
        out.append("bn%d: %.0f" % (bs, 4 * n / ms / 1e6))
    for rows, cols in ((1 << 17, 512), (1 << 19, 128), (1 << 15, 2048)):
        ms = timed(lambda: hip.quantize_rows_(x, y, rows, f, r, 448.0))
        out.append("pc%dx%d: %.0f" % (rows, cols, 4 * n / ms / 1e6))
    print(os.path.basename(hip._SO), mode, "  ".join(out), flush=True)
