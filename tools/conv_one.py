"""One convolution kind of one tools/conv_bench.py shape, N calls back to back (for a kernel trace):
conv_one.py SHAPE fp|bp|wu [N]."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from conv_bench import LAYERS, hip  # noqa: E402

shape, kind = int(sys.argv[1]), sys.argv[2]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
N, C, H, W, K, R, S, st, pd = LAYERS[shape]
OH, OW = (H + 2 * pd - R) // st + 1, (W + 2 * pd - S) // st + 1
x = torch.randn(N, C, H, W, device="cuda")
w = torch.randn(K, C, R, S, device="cuda")
b = torch.randn(K, device="cuda")
go = torch.randn(N, K, OH, OW, device="cuda")
y, gi, gw = torch.empty_like(go), torch.empty_like(x), torch.empty_like(w)
fn = {"fp": lambda: hip.conv2d_fp_(y, x, w, b, st, pd), "bp": lambda: hip.conv2d_bp_(gi, go, w, st, pd),
      "wu": lambda: hip.conv2d_wu_(gw, go, x, st, pd)}[kind]
for _ in range(reps):
    fn()
torch.cuda.synchronize()
print("shape", LAYERS[shape], kind, reps, "calls")
