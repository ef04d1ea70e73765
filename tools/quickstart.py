"""README's "Use" section as a runnable script (one MI355X): the operator, the scale policies, PTQ hooks, a training step."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fp8-emulation-toolkit_amd"))
from mpemu import mpt_emu, qutils  # noqa: E402
from mpemu.pytquant import fpemu_cuda  # noqa: E402

x = torch.randn(4096, 4096, device="cuda")
y = fpemu_cuda.FPEmuOp.apply(x, "E4M3_RNE", False, 1.0)
print("E4M3_RNE: %d distinct values" % y.unique().numel())
y = qutils.fpemu_device_fn(x, "E5M2_STOCHASTIC", inplace=False, scale=1.0)
print("E5M2_STOCHASTIC: mean error %.2e" % float((y - x).mean()))
q = qutils.quantize_tensor(x, qutils.TensorQuantConfig("e4m3", "rne", "per-tensor-max"), None)
print("per-tensor-max: max |q| %.4f for max |x| %.4f" % (float(q.abs().max()), float(x.abs().max())))


def net():
    return torch.nn.Sequential(torch.nn.Conv2d(3, 64, 3, padding=1), torch.nn.ReLU(), torch.nn.Flatten(),
                               torch.nn.Linear(64 * 16 * 16, 10)).cuda()


model, emu = mpt_emu.quantize_model(net().eval(), dtype="e4m3", device="cuda")
with torch.no_grad():
    print("PTQ forward:", tuple(model(torch.randn(8, 3, 16, 16, device="cuda")).shape))

model = net().train()
optimizer = torch.optim.SGD(model.parameters(), lr=0.01)
model, emu = mpt_emu.initialize(model, optimizer, training_algo="hybrid", device="cuda")
loss = model(torch.randn(8, 3, 16, 16, device="cuda")).square().mean()
loss.backward()
emu.optimizer_step(optimizer)
print("hybrid training step: loss %.4f" % float(loss.detach()))
