"""BERT-base 8 x 128 forward, plain or with E4M3 PTQ hooks (argv[1] = plain | hooked), 23 forwards: run under
rocprofv3 --kernel-trace --stats to split wall time into GPU kernel time and host gaps."""
import os
import sys
import time

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "fp8-emulation-toolkit_amd"))
from transformers import BertConfig, BertModel  # noqa: E402

from mpemu import mpt_emu  # noqa: E402

torch.manual_seed(0)
model = BertModel(BertConfig()).cuda().eval()
ids = torch.randint(0, 30000, (8, 128), device="cuda")
if sys.argv[1] == "hooked":
    model, emu = mpt_emu.quantize_model(model, dtype="e4m3", device="cuda")
with torch.no_grad():
    for _ in range(3):
        model(ids)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        model(ids)
    torch.cuda.synchronize()
print("%s: %.3f ms per forward (wall)" % (sys.argv[1], (time.perf_counter() - t0) / 20 * 1e3))
