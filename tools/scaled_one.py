"""N back-to-back per-tensor quantisations (amax -> scale -> cast) of one fp32 tensor, for a kernel trace.
usage: scaled_one.py MiB [entry]   entry: ws (default) | static"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fp8-emulation-toolkit_amd"))
from mpemu.pytquant.hip import _binding as hip  # noqa: E402

mib = float(sys.argv[1]) if len(sys.argv) > 1 else 1024
entry = sys.argv[2] if len(sys.argv) > 2 else "ws"
n = int(mib * (1 << 20)) // 4
x = torch.randn(n, device="cuda")
y = torch.empty_like(x)
f, r = hip.parse_mode("E4M3_RNE")
ws = torch.zeros(257, dtype=torch.int32, device="cuda")
L = hip.lib()
s = torch.cuda.current_stream().cuda_stream


def once():
    if entry == "static":
        rc = L.fpemu_quantize_per_tensor_static(x.data_ptr(), y.data_ptr(), n, 0, f, r, 448.0, 1, None, 0, 0, s)
        assert rc == 0, L.fpemu_last_error()
    else:
        hip.quantize_per_tensor_(x, y, f, r, 448.0, amax_ws=ws)


for _ in range(5):
    once()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    once()
e1.record()
torch.cuda.synchronize()
print("%s %.0f MiB: %.4f ms per call" % (entry, mib, e0.elapsed_time(e1) / 20))
