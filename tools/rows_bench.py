"""Block norm / per-row quantisation throughput by block size (64 Mi elements; 8 B/elem fp32, 4 B/elem 16-bit
while the block is register-resident, i.e. up to 8192 fp32 / 16384 16-bit elements)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fp8-emulation-toolkit_amd"))
from mpemu.pytquant.hip import _binding as hip  # noqa: E402

hip._SO = os.environ.get("FPEMU_LIB") or hip._SO  # tuning aid: a library built with other geometry constants

n = 1 << 26


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for dt, bpe in ((torch.float32, 8), (torch.bfloat16, 4)):
    x = (torch.randn(n, device="cuda") * 3).to(dt)
    y = torch.empty_like(x)
    for mode in ("E4M3_RNE", "E5M2_STOCHASTIC"):
        f, r = hip.parse_mode(mode)
        for bs in (1, 4, 8, 16, 32, 64, 128, 256, 1024, 4096, 8192, 16384, 32768, 65536):
            ms = timed(lambda: hip.quantize_blocks_(x, y, bs, f, r))
            print("%-9s %-16s block %6d  %.4f ms  %7.1f GB/s" % (str(dt).split(".")[1], mode, bs, ms,
                                                                  bpe * n / ms / 1e6), flush=True)

# LLM-shaped weights through the per-channel entry point (rows do not divide 64 Mi: own buffers)
for rows, cols in ((4096, 11008), (4096, 14336), (11008, 4096), (32000, 4096)):
    for dt, bpe in ((torch.float32, 8), (torch.bfloat16, 4)):
        x = (torch.randn(rows, cols, device="cuda") * 3).to(dt)
        y = torch.empty_like(x)
        f, r = hip.parse_mode("E4M3_RNE")
        ms = timed(lambda: hip.quantize_rows_(x, y, rows, f, r, 448.0))
        print("%-9s E4M3_RNE per-channel %6d x %6d  %.4f ms  %7.1f GB/s" % (str(dt).split(".")[1], rows, cols, ms,
                                                                          bpe * rows * cols / ms / 1e6), flush=True)
