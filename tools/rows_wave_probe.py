"""Tuning aid: per-channel rows of 1280 ... 8192 elements, a WAVE per row (FPEMU_ROWS_WAVE_NV = 8 / 16 vectors per lane)
against the shipped choice (a wave up to 4 vectors per lane, the 256-thread block above).  FPEMU_LIB selects a build."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fp8-emulation-toolkit_amd"))
from mpemu.pytquant.hip import _binding as hip  # noqa: E402

hip._SO = os.environ.get("FPEMU_LIB") or hip._SO
n = 1 << 26


def timed(fn, reps=40):
    for _ in range(8):
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
f, r = hip.parse_mode(os.environ.get("MODE", "E4M3_RNE"))
for dt, bpe, lens in ((torch.float32, 8, (1024, 1280, 2048, 3072, 4096, 8192)), (torch.bfloat16, 4, (2048, 4096, 8192, 16384))):
    x = (torch.randn(n, device="cuda") * 3).to(dt)
    y = torch.empty_like(x)
    for cols in lens:
        rows = n // cols
        xx, yy = x[: rows * cols], y[: rows * cols]
        ms = min(timed(lambda: hip.quantize_rows_(xx, yy, rows, f, r, 448.0)) for _ in range(3))
        out.append("%s/%d: %.0f" % (str(dt).split(".")[1][:4], cols, bpe * rows * cols / ms / 1e6))
print("WAVE_NV=%s %s | %s" % (os.environ.get("FPEMU_ROWS_WAVE_NV", "4"), os.path.basename(hip._SO), "  ".join(out)), flush=True)
