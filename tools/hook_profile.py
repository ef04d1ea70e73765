"""Where the host time of a hooked forward goes: cProfile over BERT-base (8 x 128) with E4M3 PTQ hooks."""
import cProfile
import os
import pstats
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
model, emu = mpt_emu.quantize_model(model, dtype="e4m3", device="cuda")
with torch.no_grad():
    for _ in range(5):
        model(ids)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        model(ids)
    torch.cuda.synchronize()
    print("hooked forward %.3f ms" % ((time.perf_counter() - t0) / 20 * 1e3))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(20):
        model(ids)
    torch.cuda.synchronize()
    pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
st.sort_stats("cumulative").print_stats("mpemu|hooks", 25)
