"""BASELINE config 3 timing: BERT-base (random init) forward, batch 8 x 128 tokens, plain vs
mpt_emu.quantize_model(dtype="e4m3") post-training quantisation hooks."""
import os
import sys
import time

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "fp8-emulation-toolkit_amd"))
from transformers import BertConfig, BertModel  # noqa: E402

from mpemu import mpt_emu  # noqa: E402


def run(model, ids, n):
    with torch.no_grad():
        for _ in range(3):
            model(ids)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            model(ids)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


torch.manual_seed(0)
model = BertModel(BertConfig()).cuda().eval()
ids = torch.randint(0, 30000, (8, 128), device="cuda")
plain = run(model, ids, 20)
from mpemu.graphs import GraphedForward  # noqa: E402

# torch's own kernels under capture vs eager, BEFORE any hook exists: the yardstick for the hooked comparison below
with torch.no_grad():
    ref_eager = model(ids).last_hidden_state.clone()
ref_graph = GraphedForward(model, ids)(ids).last_hidden_state.clone()
unhooked_diff = float((ref_eager - ref_graph).abs().max())
model, emu = mpt_emu.quantize_model(model, dtype="e4m3", device="cuda")
quant = run(model, ids, 20)
fwd = GraphedForward(model, ids)
with torch.no_grad():
    want = model(ids).last_hidden_state.clone()
got = fwd(ids).last_hidden_state
same = bool(torch.equal(want, got))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    fwd(ids)
torch.cuda.synchronize()
graphed = (time.perf_counter() - t0) / 20 * 1e3
print("BERT-base forward 8x128: plain %.2f ms, E4M3 PTQ hooks %.2f ms (x%.2f), hooked forward as one HIP graph "
      "%.2f ms; graph vs eager: hooked same bits %s (max diff %.3g), UNHOOKED model max diff %.3g -- torch's own kernels "
      "differ under capture before any hook exists, and a cast turns a last-bit difference into a whole code step; "
      "hooked modules alone replay bit-equal (tests/test_gpu_graph.py); modules %d" % (
          plain, quant, quant / plain, graphed, same, float((want - got).abs().max()), unhooked_diff,
          len(emu.emulator.model_qconfig_dict)))
