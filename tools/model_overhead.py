"""End-to-end cost of the emulation hooks: ResNet-50 training step (batch 32, 224x224, fp32 compute) with and
without mpt_emu.initialize(training_algo="hybrid").  Reports ms/step and the number of fpemu launches."""
import os
import sys
import time

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "fp8-emulation-toolkit_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _resnet import ResNet50  # noqa: E402

from mpemu import mpt_emu  # noqa: E402


def steps(model, opt, x, t, emu, n):
    for _ in range(n):
        opt.zero_grad(set_to_none=True)
        loss = torch.nn.functional.cross_entropy(model(x), t)
        loss.backward()
        if emu is None:
            opt.step()
        else:
            emu.optimizer_step(opt)
    torch.cuda.synchronize()


def run(emulate, batch=32):
    torch.manual_seed(0)
    model = ResNet50().cuda().train()
    opt = torch.optim.SGD(model.parameters(), lr=0.01, momentum=0.9)
    emu = None
    if emulate:
        model, emu = mpt_emu.initialize(model, opt, training_algo="hybrid", device="cuda")
    x = torch.randn(batch, 3, 224, 224, device="cuda")
    t = torch.randint(0, 1000, (batch,), device="cuda")
    steps(model, opt, x, t, emu, 3)
    t0 = time.perf_counter()
    steps(model, opt, x, t, emu, 10)
    return (time.perf_counter() - t0) / 10 * 1e3


if __name__ == "__main__":
    if "emu-only" in sys.argv:
        print("hybrid FP8 emulation %.1f ms" % run(True))
        sys.exit(0)
    base = run(False)
    emul = run(True)
    print("ResNet-50 batch 32 fp32 training step: plain %.1f ms, hybrid FP8 emulation %.1f ms (x%.2f)" % (
        base, emul, emul / base))
