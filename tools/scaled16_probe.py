"""The per-tensor scaled path by storage type: quantize_tensor (amax + device-side scale + cast) and the amax pass alone."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fp8-emulation-toolkit_amd"))
from mpemu import qutils
from mpemu.pytquant.hip import _binding as hip
cfg = qutils.TensorQuantConfig("e4m3", "rne", "per-tensor")
def timed(f, n=20):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for dt, b in ((torch.float32, 12), (torch.bfloat16, 6), (torch.float16, 6)):
    for n in (1 << 24, 1 << 27):
        x = (torch.randn(n, device="cuda") * 3).to(dt)
        ms = timed(lambda: qutils.quantize_tensor(x, cfg, None, inplace=False))
        w = torch.zeros(1, dtype=torch.int32, device="cuda")
        ms_a = timed(lambda: hip.amax_(x, w))
        print("%-9s %10d  quantize_tensor per-tensor %.4f ms  %.0f GB/s (%d B/elem)   amax alone %.4f ms %.0f GB/s" % (
            str(dt).split(".")[1], n, ms, b * n / ms / 1e6, b, ms_a, (b // 3) * n / ms_a / 1e6), flush=True)
