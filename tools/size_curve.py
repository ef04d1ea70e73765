"""Launch time against size: the flat cast (E4M3_RNE, fp32, unscaled and from a device amax word), the bare copy of
tools/ceilings.hip and torch's own elementwise kernel over the same buffers -- back-to-back launches between two
events, so every leg carries the same launch boundary.  Fits t = a + bytes / rate over the sizes of each leg: `a` is
what a launch costs before and after it streams."""
import ctypes
import os
import sys

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "fp8-emulation-toolkit_amd"))
from mpemu.pytquant.hip import _binding as hip  # noqa: E402

hip._SO = os.environ.get("FPEMU_LIB") or hip._SO
C = ctypes.CDLL(os.path.join(ROOT, "tools", "libfpemu_ceilings.so"))
C.ceil_copy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_void_p]
stream = torch.cuda.current_stream().cuda_stream
f, r = hip.parse_mode("E4M3_RNE")
fs, rs = hip.parse_mode("E5M2_STOCHASTIC")


def timed(fn, launches):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(launches):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / launches * 1e3   # us


sizes = [1 << k for k in range(14, 29, 2)]    # fp32 elements: 64 KiB ... 1 GiB of input
word = torch.zeros(1, dtype=torch.int32, device="cuda")
legs = {}
print("%12s %10s %10s %10s %10s %10s" % ("elements", "cast", "cast(word)", "cast SR", "bare copy", "torch mul"))
for n in sizes:
    x = torch.randn(n, device="cuda")
    y = torch.empty_like(x)
    word.fill_(0x40400000)    # bits(3.0f): any finite maximum will do for a rate
    launches = max(20, min(400, (1 << 31) // (8 * n)))
    row = {
        "cast": timed(lambda: hip.cast_(x, y, f, r, scale=1.0), launches),
        "cast(word)": timed(lambda: hip.cast_scaled_(x, y, f, r, word, 448.0), launches),
        "cast SR": timed(lambda: hip.cast_(x, y, fs, rs, scale=1.0, seed=1, offset=2), launches),
        "bare copy": timed(lambda: C.ceil_copy(x.data_ptr(), y.data_ptr(), n // 4, 128, stream), launches),
        "torch mul": timed(lambda: torch.mul(x, 2.0, out=y), launches),
    }
    for k, v in row.items():
        legs.setdefault(k, []).append((8.0 * n, v))
    print("%12d %10.2f %10.2f %10.2f %10.2f %10.2f" % (n, row["cast"], row["cast(word)"], row["cast SR"],
                                                      row["bare copy"], row["torch mul"]), flush=True)

print("fit t = a + bytes / rate over the four largest sizes:")
for k, pts in legs.items():
    pts = pts[-4:]
    mx = sum(p[0] for p in pts) / len(pts)
    my = sum(p[1] for p in pts) / len(pts)
    slope = sum((p[0] - mx) * (p[1] - my) for p in pts) / sum((p[0] - mx) ** 2 for p in pts)
    print("  %-11s a = %6.2f us   rate = %7.1f GB/s   (smallest size: %.2f us per launch)" % (
        k, my - slope * mx, 1e-3 / slope, legs[k][0][1]))
