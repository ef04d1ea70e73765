"""Tuning aid: per-workgroup timeline of the implicit convolution's GEMM (needs a library built with
-DFPEMU_GEMM_TRACE, passed as FPEMU_LIB).  argv: layer index of tools/conv_bench.py's LAYERS (default 1)."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fp8-emulation-toolkit_amd"))
from mpemu.pytquant.hip import _binding as hip  # noqa: E402

hip._SO = os.environ.get("FPEMU_LIB") or hip._SO
LAYERS = [(32, 3, 224, 224, 64, 7, 7, 2, 3), (32, 64, 56, 56, 64, 3, 3, 1, 1), (32, 128, 28, 28, 128, 3, 3, 1, 1),
          (32, 256, 14, 14, 256, 3, 3, 1, 1), (32, 512, 7, 7, 512, 3, 3, 1, 1), (32, 256, 56, 56, 64, 1, 1, 1, 0)]
N, C, H, W, K, R, S, st, pd = LAYERS[int(sys.argv[1]) if len(sys.argv) > 1 else 1]
OH, OW = (H + 2 * pd - R) // st + 1, (W + 2 * pd - S) // st + 1
x = torch.randn(N, C, H, W, device="cuda")
w = torch.randn(K, C, R, S, device="cuda")
b = torch.randn(K, device="cuda")
y = torch.empty(N, K, OH, OW, device="cuda")
L = hip.lib()
L.fpemu_test_conv_trace_.argtypes = [ctypes.c_void_p]
L.fpemu_test_conv_trace_.restype = ctypes.c_int
for _ in range(3):
    hip.conv2d_fp_(y, x, w, b, st, pd)
torch.cuda.synchronize()
trace = torch.zeros(3 * 65536, dtype=torch.int64, device="cuda")
phase = torch.zeros(4 * 65536, dtype=torch.int64, device="cuda")
L.fpemu_test_conv_phase_.argtypes = [ctypes.c_void_p]
L.fpemu_test_conv_phase_.restype = ctypes.c_int
assert L.fpemu_test_conv_trace_(trace.data_ptr()) == 0
assert L.fpemu_test_conv_phase_(phase.data_ptr()) == 0
hip.conv2d_fp_(y, x, w, b, st, pd)
torch.cuda.synchronize()
L.fpemu_test_conv_trace_(None)
L.fpemu_test_conv_phase_(None)
ph = phase.cpu().numpy().reshape(-1, 4)
ph = ph[ph[:, 0] != 0].astype(np.float64) / 100.0
first = ph[:, 0] < ph[:, 0].min() + 10.0
for name, sel in (("first round (started within 10 us)", first), ("later rounds", ~first)):
    if sel.any():
        q = ph[sel]
        print("%s: %d workgroups; prologue (entry -> first tile in LDS) %.1f us, k loop %.1f us, epilogue %.1f us (medians)" % (
            name, sel.sum(), np.median(q[:, 1] - q[:, 0]), np.median(q[:, 2] - q[:, 1]), np.median(q[:, 3] - q[:, 2])))
t = trace.cpu().numpy().reshape(-1, 3)
t = t[t[:, 0] != 0]
t0, t1, hw = t[:, 0], t[:, 1], t[:, 2]
base = t0.min()
s, e = (t0 - base) / 100.0, (t1 - base) / 100.0   # us (100 MHz)
xcc = (hw >> 32) & 0xF
cu = (hw >> 8) & 0xF
sh = (hw >> 12) & 0x1
se = (hw >> 13) & 0x7
cuid = xcc * 1000 + se * 100 + sh * 10 + cu
print("workgroups %d, span %.1f us, duration min/median/max %.1f / %.1f / %.1f us" % (len(t), e.max(), (e - s).min(), np.median(e - s), (e - s).max()))
print("distinct CUs seen:", len(np.unique(cuid)), " per XCC workgroups:", np.bincount(xcc.astype(int)))
for lo in np.arange(0, e.max(), e.max() / 16):
    hi = lo + e.max() / 16
    running = ((s < hi) & (e > lo)).sum()
    started = ((s >= lo) & (s < hi)).sum()
    print("  %.1f-%.1f us: resident %4d  started %4d  median duration of those started %.1f" % (lo, hi, running, started, np.median((e - s)[(s >= lo) & (s < hi)]) if started else 0))
per_cu = np.array([((cuid == c).sum()) for c in np.unique(cuid)])
print("workgroups per CU: min %d max %d" % (per_cu.min(), per_cu.max()), np.bincount(per_cu))
last = np.array([e[cuid == c].max() for c in np.unique(cuid)])
print("CU finish time: min %.1f median %.1f max %.1f us" % (last.min(), np.median(last), last.max()))
