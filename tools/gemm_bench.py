"""Time the cmodel GEMM kernels (both variants) on the GPU: TFLOP/s vs their rooflines."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fp8-emulation-toolkit_amd"))
from mpemu.pytquant.hip import _binding as hip  # noqa: E402

hip._SO = os.environ.get("FPEMU_LIB") or hip._SO  # tuning aid: a library built with other geometry constants

PEAK = {1: 157.3, 0: 78.6}


ITERS = int(os.environ.get("GEMM_ITERS", "0"))   # 0: as many launches as fill ~80 ms (the clock ramp of the first launches
                                                 # after an idle GPU reads 10-15 % low on a 1 ms kernel timed over 5 launches)


def run(M, N, K, at, bt, variant, iters=None):
    iters = iters or ITERS
    A = torch.randn((K, M) if at else (M, K), device="cuda")
    B = torch.randn((N, K) if bt else (K, N), device="cuda")
    C = torch.zeros(M, N, device="cuda")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    hip.gemm_(C, A, B, 1.0, at, bt, variant)
    e0.record()
    hip.gemm_(C, A, B, 1.0, at, bt, variant)
    e1.record()
    torch.cuda.synchronize()
    if not iters:
        iters = max(5, min(200, int(80.0 / max(e0.elapsed_time(e1), 1e-3)) + 1))
    for _ in range(1 + iters // 5):
        hip.gemm_(C, A, B, 1.0, at, bt, variant)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        hip.gemm_(C, A, B, 1.0, at, bt, variant)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    tf = 2.0 * M * N * K / ms / 1e9
    print("variant=%d %5dx%5dx%5d at=%d bt=%d  %8.3f ms  %7.2f TFLOP/s  frac=%.3f" % (
        variant, M, N, K, at, bt, ms, tf, tf / PEAK[variant]), flush=True)


def run_chain(M, N, K, variant=1, iters=20):
    """fpemu_gemm_chain_f32 (accumulators start at C, C = acc): the CHAIN instantiation of the same kernels."""
    A = torch.randn(M, K, device="cuda")
    B = torch.randn(K, N, device="cuda")
    C = torch.zeros(M, N, device="cuda")
    st = torch.cuda.current_stream().cuda_stream

    def go():
        rc = hip.lib().fpemu_gemm_chain_f32(C.data_ptr(), A.data_ptr(), B.data_ptr(), M, N, K, K, N, N, 0, 0, variant,
                                            1, 0, 0, 0, 1, 0, 0, st)
        assert rc == 0, rc
    for _ in range(5):
        go()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        go()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    tf = 2.0 * M * N * K / ms / 1e9
    print("chain variant=%d %5dx%5dx%5d  %8.3f ms  %7.2f TFLOP/s  frac=%.3f" % (variant, M, N, K, ms, tf, tf / PEAK[variant]),
          flush=True)


if __name__ == "__main__":
    sizes = [(4096, 4096, 4096), (8192, 8192, 8192)]
    args = [a for a in sys.argv[1:] if "x" in a]
    if args:
        sizes = [tuple(int(v) for v in s.split("x")) for s in args]
    variants = (1,) if "v1" in sys.argv else (0,) if "v0" in sys.argv else (1, 0)
    trans = ((False, False),) if "nn" in sys.argv else ((False, False), (False, True), (True, False), (True, True))
    for v in variants:
        for (M, N, K) in sizes:
            if "chain" in sys.argv:
                run_chain(M, N, K, v)
                continue
            for at, bt in trans:
                run(M, N, K, at, bt, v)
