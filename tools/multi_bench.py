"""quantize_model_weights-shaped work: per-tensor E4M3 quantisation of a model's weights, the loop of single-tensor
calls against fpemu_quantize_per_tensor_multi.  Wall time per pass (host + device, synchronised at the end)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fp8-emulation-toolkit_amd"))
from mpemu.pytquant.hip import _binding as hip  # noqa: E402

f, r = hip.parse_mode("E4M3_RNE")
dev = "cuda:0"
shapes = {
    "bert-base (73 Linear weights)": [(768, 768)] * 48 + [(3072, 768)] * 12 + [(768, 3072)] * 12 + [(768, 768)],
    "resnet-50 (54 conv / fc weights)": [(64, 3, 7, 7)] + [(64, 64, 3, 3)] * 3 + [(256, 64, 1, 1)] * 7 + [(128, 128, 3, 3)] * 4
    + [(512, 128, 1, 1)] * 9 + [(256, 256, 3, 3)] * 6 + [(1024, 256, 1, 1)] * 13 + [(512, 512, 3, 3)] * 3 + [(2048, 512, 1, 1)] * 7
    + [(1000, 2048)],
}
for name, shp in shapes.items():
    ws = [torch.randn(*s, device=dev) for s in shp]
    outs = [torch.empty_like(w) for w in ws]

    def loop():
        for w, o in zip(ws, outs):
            hip.quantize_per_tensor_(w, o, f, r, 448.0)

    def multi():
        hip.quantize_per_tensor_multi_(ws, f, r, 448.0, dsts=outs2, ws=scratch)

    outs2 = [torch.empty_like(w) for w in ws]
    scratch = torch.empty(len(ws) + 32 * 256, dtype=torch.int32, device=dev)
    res = {}
    for label, fn in (("loop", loop), ("multi", multi)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        res[label] = (time.perf_counter() - t0) / 20 * 1e3
    same = all(torch.equal(a, b) for a, b in zip(outs, outs2))
    mb = sum(w.numel() for w in ws) * 4 / 2 ** 20
    print("%s, %.0f MiB: loop %.3f ms  multi %.3f ms  (x%.2f)  same bits: %s" % (name, mb, res["loop"], res["multi"], res["loop"] / res["multi"], same), flush=True)
