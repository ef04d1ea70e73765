"""Time the cmodel convolution (fp / bp / wu, MFMA fmaf-chain variant) on ResNet-style layers."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fp8-emulation-toolkit_amd"))
from mpemu.pytquant.hip import _binding as hip  # noqa: E402

hip._SO = os.environ.get("FPEMU_LIB") or hip._SO  # tuning aid: a library built with other constants

LAYERS = [(32, 3, 224, 224, 64, 7, 7, 2, 3), (32, 64, 56, 56, 64, 3, 3, 1, 1), (32, 128, 28, 28, 128, 3, 3, 1, 1),
          (32, 256, 14, 14, 256, 3, 3, 1, 1), (32, 512, 7, 7, 512, 3, 3, 1, 1), (32, 256, 56, 56, 64, 1, 1, 1, 0),
          # ResNet-50's strided layers (the gathered form): 3 x 3 stride 2, 1 x 1 stride 2 (downsample)
          (32, 128, 56, 56, 128, 3, 3, 2, 1), (32, 256, 28, 28, 256, 3, 3, 2, 1), (32, 512, 14, 14, 512, 3, 3, 2, 1),
          (32, 256, 56, 56, 512, 1, 1, 2, 0), (32, 512, 28, 28, 1024, 1, 1, 2, 0), (32, 1024, 14, 14, 2048, 1, 1, 2, 0)]


def timed(fn, iters=3):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


if __name__ == "__main__":
    sel = [int(a) for a in sys.argv[1:]]
    for (N, C, H, W, K, R, S, st, pd) in ([LAYERS[i] for i in sel] if sel else LAYERS):
        OH, OW = (H + 2 * pd - R) // st + 1, (W + 2 * pd - S) // st + 1
        x = torch.randn(N, C, H, W, device="cuda")
        w = torch.randn(K, C, R, S, device="cuda")
        b = torch.randn(K, device="cuda")
        go = torch.randn(N, K, OH, OW, device="cuda")
        y, gi, gw = torch.empty_like(go), torch.empty_like(x), torch.empty_like(w)
        flop = 2.0 * N * K * C * R * S * OH * OW
        t_fp = timed(lambda: hip.conv2d_fp_(y, x, w, b, st, pd))
        t_bp = timed(lambda: hip.conv2d_bp_(gi, go, w, st, pd))
        t_wu = timed(lambda: hip.conv2d_wu_(gw, go, x, st, pd))
        print("N%d C%d %dx%d K%d %dx%d s%d p%d  %.2f GFLOP | fp %.3f ms (%.1f TF) | bp %.3f ms (%.1f TF) | wu %.3f ms (%.1f TF)" % (
            N, C, H, W, K, R, S, st, pd, flop / 1e9, t_fp, flop / t_fp / 1e9, t_bp, flop / t_bp / 1e9, t_wu,
            flop / t_wu / 1e9), flush=True)
