"""Cast throughput by storage dtype and scale (128 Mi elements), then the same BYTES three ways: the 16-bit cast,
the fp32 cast over half the elements and the bare copy of tools/ceilings.hip -- a 16-bit launch moves half the
bytes of an fp32 one, so the fixed ramp-up / drain of a launch weighs twice as much in its rate."""
import ctypes
import os
import sys

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "fp8-emulation-toolkit_amd"))
from mpemu.pytquant.hip import _binding as hip  # noqa: E402

hip._SO = os.environ.get("FPEMU_LIB", hip._SO)  # tuning aid: a library built with other geometry constants


def timed(f, launches=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(launches):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / launches


n = 1 << 27
for dt, bpe in ((torch.float32, 8), (torch.float16, 4), (torch.bfloat16, 4), (torch.float64, 16)):
    x = (torch.randn(n, device="cuda") * 3).to(dt)
    y = torch.empty_like(x)
    for mode in os.environ.get("MODES", "E4M3_RNE E5M2_STOCHASTIC").split():
        f, r = hip.parse_mode(mode)
        for scale in (1.0, 3.0):
            ms = timed(lambda: hip.cast_(x, y, f, r, scale=scale, seed=1, offset=2))
            print("%-9s %-16s scale %.0f  %.4f ms  %7.1f GB/s" % (str(dt).split(".")[1], mode, scale, ms,
                                                                  bpe * n / ms / 1e6), flush=True)
del x, y

so = os.path.join(ROOT, "tools", "libfpemu_ceilings.so")
if os.path.isfile(so):
    C = ctypes.CDLL(so)
    C.ceil_copy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_void_p]
    stream = torch.cuda.current_stream().cuda_stream
    f, r = hip.parse_mode("E4M3_RNE")
    print("same bytes per launch, three ways (E4M3_RNE, scale 3; GB/s read + written):")
    for mib in (128, 256, 512, 1024, 2048):
        byts = mib << 20                      # read + written per launch
        xb = (torch.randn(byts // 4, device="cuda") * 3).to(torch.bfloat16)   # byts / 2 bytes of bf16
        yb = torch.empty_like(xb)
        xf = xb.view(torch.float32)           # the same bytes as fp32 (any bit patterns will do for a rate)
        yf = torch.empty_like(xf)
        ms16 = timed(lambda: hip.cast_(xb, yb, f, r, scale=3.0))
        ms32 = timed(lambda: hip.cast_(xf, yf, f, r, scale=3.0))
        msc = min(timed(lambda: C.ceil_copy(xf.data_ptr(), yf.data_ptr(), xf.numel() // 4, b, stream))
                  for b in (128, 256))
        print("  %5d MiB  bf16 cast %.4f ms %7.1f | fp32 cast %.4f ms %7.1f | bare copy %.4f ms %7.1f" % (
            mib, ms16, byts / ms16 / 1e6, ms32, byts / ms32 / 1e6, msc, byts / msc / 1e6), flush=True)
