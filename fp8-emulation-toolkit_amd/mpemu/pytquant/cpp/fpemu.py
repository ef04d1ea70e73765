"""Placeholder for the reference's CPU operator module (mpemu/pytquant/cpp/fpemu.py).

This build is MI355X-only: there is deliberately no CPU implementation of the cast in the
product (the CPU restatement under oracle/ is test infrastructure).  The module exists so that
`from mpemu.pytquant import fpemu_cpp` keeps importing; using it raises.
"""
from torch.autograd import Function


class FPEmuOp(Function):
    @staticmethod
    def forward(ctx, input, mode='NONE', inplace=False, scale=1.0, blocknorm=False, blocksize=1):
        if mode == 'NONE':
            ctx.mark_dirty(input)
            return input
        raise RuntimeError(
            "mpemu (MI355X build): the fpemu cast runs on ROCm devices only; move the tensor to the GPU "
            "(got a tensor on %s)" % (input.device,))

    @staticmethod
    def backward(ctx, output_grad):
        return output_grad, None, None, None, None, None
