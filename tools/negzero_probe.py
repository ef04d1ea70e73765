"""Probe: chains of signed-zero products through every chain-GEMM kernel (tile sizes by problem size)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fp8-emulation-toolkit_amd"))
from mpemu.pytquant.hip import _binding as hip  # noqa: E402

L = hip.lib()
for variant in (1, 0):
    for (M, N, K) in ((16, 16, 64), (64, 16, 576), (256, 256, 64), (512, 512, 64), (1024, 1024, 64), (4096, 4096, 64),
                      (48, 20, 130)):
        for (a, b, c0, want) in ((-1.0, 0.0, -0.0, True), (1.0, 0.0, -0.0, False), (-1.0, 0.0, 0.0, False)):
            A = torch.full((M, K), a, device="cuda")
            B = torch.full((K, N), b, device="cuda")
            C = torch.full((M, N), c0, device="cuda")
            rc = L.fpemu_gemm_chain_f32(C.data_ptr(), A.data_ptr(), B.data_ptr(), M, N, K, K, N, N, 0, 0, variant, 1, 0, 0,
                                        0, 1, 0, 0, None)
            torch.cuda.synchronize()
            neg = torch.signbit(C)
            ok = bool(neg.all()) if want else not bool(neg.any())
            print("variant %d %5dx%5dx%4d  a=%+.0f b=%+.0f c=%s  -> %s (%d of %d negative)" % (
                variant, M, N, K, a, b, "-0" if str(c0).startswith("-") else "+0", "ok" if ok else "WRONG",
                int(neg.sum()), neg.numel()))
