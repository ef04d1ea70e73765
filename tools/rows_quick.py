"""Tuning aid: the long-row cases of tools/rows_bench.py only (FPEMU_LIB selects a library build)."""
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


out = []
for dt, bpe in ((torch.float32, 8), (torch.bfloat16, 4)):
    x = (torch.randn(n, device="cuda") * 3).to(dt)
    y = torch.empty_like(x)
    f, r = hip.parse_mode("E4M3_RNE")
    for bs in (4096, 8192, 16384, 32768, 65536):
        ms = timed(lambda: hip.quantize_blocks_(x, y, bs, f, r))
        out.append("%s/%d: %.0f" % (str(dt).split(".")[1][:4], bs, bpe * n / ms / 1e6))
    for rows, cols in ((4096, 11008), (4096, 14336)):
        xx, yy = x[: rows * cols], y[: rows * cols]
        ms = timed(lambda: hip.quantize_rows_(xx, yy, rows, f, r, 448.0))
        out.append("%s/%dx%d: %.0f" % (str(dt).split(".")[1][:4], rows, cols, bpe * rows * cols / ms / 1e6))
print(os.path.basename(hip._SO), "  ".join(out), flush=True)
