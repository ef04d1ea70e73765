"""Output quantisation after the c-model GEMM / conv: amax pass + cast (fpemu_quantize_per_tensor_ws) against the cast
alone from the maximum the op's epilogue found (fpemu_gemm_f32_amax / fpemu_conv2d_fp_f32_amax + fpemu_cast_scaled),
and what the epilogue costs the op itself."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fp8-emulation-toolkit_amd"))
from mpemu.pytquant.hip import _binding as hip  # noqa: E402


def timed(fn, reps, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


f, r = hip.parse_mode("E4M3_RNE")
word = torch.zeros(1, dtype=torch.int32, device="cuda")
ws = torch.zeros(257, dtype=torch.int32, device="cuda")
for (M, N, K) in ((8192, 8192, 8192), (4096, 4096, 4096), (2048, 8192, 1024)):
    A = torch.randn(M, K, device="cuda")
    B = torch.randn(K, N, device="cuda")
    C = torch.zeros(M, N, device="cuda")
    Q = torch.empty_like(C)
    for variant in (1, 0):
        reps = 5 if M * N * K > 2 ** 36 else 20
        t_plain = timed(lambda: hip.gemm_(C, A, B, 1.0, False, False, variant), reps)
        t_amax = timed(lambda: hip.gemm_(C, A, B, 1.0, False, False, variant, amax=word), reps)
        print("gemm %dx%dx%d variant %d: plain %.4f ms | with epilogue amax %.4f ms (%+.2f %%)" % (
            M, N, K, variant, t_plain, t_amax, 100 * (t_amax / t_plain - 1)), flush=True)
    C.normal_()
    word.zero_()
    hip.amax_(C, word)
    t_two = timed(lambda: hip.quantize_per_tensor_(C, Q, f, r, 448.0, amax_ws=ws), 50)
    t_one = timed(lambda: hip.cast_scaled_(C, Q, f, r, word, 448.0), 50)
    n = C.numel()
    print("output %d MiB: amax + cast %.4f ms (%.0f GB/s at 12 B/elem) | cast from the epilogue's word %.4f ms (%.0f GB/s at 8 B/elem)"
          % (n * 4 >> 20, t_two, 12 * n / t_two / 1e6, t_one, 8 * n / t_one / 1e6), flush=True)
    del A, B, C, Q

x = torch.randn(32, 64, 56, 56, device="cuda")
w = torch.randn(64, 64, 3, 3, device="cuda")
y = torch.empty(32, 64, 56, 56, device="cuda")
q = torch.empty_like(y)
t_plain = timed(lambda: hip.conv2d_fp_(y, x, w, None, 1, 1, 1), 50)
t_amax = timed(lambda: hip.conv2d_fp_(y, x, w, None, 1, 1, 1, amax=word), 50)
t_two = timed(lambda: hip.quantize_per_tensor_(y, q, f, r, 448.0, amax_ws=ws), 100)
t_one = timed(lambda: hip.cast_scaled_(y, q, f, r, word, 448.0), 100)
print("conv fp 32x64x56x56 3x3: plain %.4f ms | with amax %.4f ms;  output quantise %.4f -> %.4f ms" % (t_plain, t_amax, t_two, t_one))
