"""Per-call cost of the Python-visible entry points on small tensors (configs 3/4 regime)."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fp8-emulation-toolkit_amd"))
import torch
from mpemu import qutils
from mpemu.pytquant.hip import _binding as hip
from mpemu.pytquant import fpemu_cuda

dev = "cuda:0"
qutils._MEMO_ON = False   # every call here repeats the last one: time the call, not the memo of its result
cfg_t = qutils.TensorQuantConfig("e4m3", "rne", "per-tensor")
cfg_c = qutils.TensorQuantConfig("e4m3", "rne", "per-channel")
cfg_n = qutils.TensorQuantConfig("e5m2", "stochastic")
fmt, rm = hip.parse_mode("E4M3_RNE")

def bench(fn, iters=300):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(iters): fn()
    e1.record(); t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_wall = time.perf_counter() - t0
    return e0.elapsed_time(e1) * 1e3 / iters, t_host * 1e6 / iters, t_wall * 1e6 / iters

print("%-34s %10s %12s %12s %12s" % ("op", "elements", "gpu us/call", "host us/call", "wall us/call"))
for shape in [(64, 64), (768, 768), (3072, 768), (4096, 4096)]:
    x = torch.randn(*shape, device=dev); o = torch.empty_like(x)
    n = x.numel()
    for name, fn in [
        ("raw cast_ (ctypes)", lambda: hip.cast_(x, o, fmt, rm)),
        ("FPEmuOp.apply E4M3_RNE", lambda: fpemu_cuda.FPEmuOp.apply(x, "E4M3_RNE", False, 1.0)),
        ("quantize_tensor per-tensor", lambda: qutils.quantize_tensor(x, cfg_t, None)),
        ("quantize_tensor per-channel", lambda: qutils.quantize_tensor(x, cfg_c, None)),
        ("quantize_tensor e5m2 SR unscaled", lambda: qutils.quantize_tensor(x, cfg_n, None)),
        ("torch ref: x.abs().max()", lambda: x.abs().max()),
        ("torch ref: x*2 (elementwise)", lambda: torch.mul(x, 2.0)),
    ]:
        g, h, w = bench(fn)
        print("%-34s %10d %12.2f %12.2f %12.2f" % (name, n, g, h, w))
