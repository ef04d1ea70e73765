"""FPEmuOp -- the autograd operator of the fpemu cast, MI355X build.

Same operator the reference exposes as mpemu.pytquant.{cpp,cuda,hip}.fpemu.FPEmuOp
(mpemu/pytquant/cpp/fpemu.py:43-71): `FPEmuOp.apply(tensor, mode, inplace=False, scale=1.0,
blocknorm=False, blocksize=1)`; mode strings as in that file's docstring; 'NONE' passes the
tensor through; sparse COO tensors are cast on their values; backward is the straight-through
estimator.

Differences from the reference, all deliberate (DESIGN.md "Boundary"):
  * an unknown mode string raises ValueError (the reference silently returns zeros / the input);
  * bfloat16 tensors are accepted (the reference GPU path rejects them); float64 tensors take plain casts
    (block norm excepted), as on the reference's GPU path;
  * backward returns one gradient per forward input (the reference returns 5 for 6 inputs);
  * importing this module does not call torch.manual_seed (the reference does, fpemu.py:19);
  * stochastic modes draw from a counter-based Philox stream: see manual_seed() below.
"""
import torch
from torch.autograd import Function

from . import _binding as fpemu_hip

manual_seed = fpemu_hip.manual_seed
get_rng_state = fpemu_hip.get_rng_state
set_rng_state = fpemu_hip.set_rng_state


class FPEmuOp(Function):
    @classmethod
    def apply(cls, input, mode='NONE', inplace=False, scale=1.0, blocknorm=False, blocksize=1):
        """Same call as the reference's FPEmuOp.apply.  A dense device tensor and a plain cast -- the case every
        hook and emulator call is -- goes through the C++ extension (checks, allocation, launch and the
        straight-through autograd node in one native call); everything else takes the Function below."""
        ext = fpemu_hip.ext()
        if (ext is not None and mode != 'NONE' and not blocknorm and input.is_cuda and not input.is_sparse
                and not fpemu_hip.tracing()):
            fmt, rmode = fpemu_hip.parse_mode(mode)
            return ext.fpemu_op(input, bool(inplace), fmt, rmode, float(scale), fpemu_hip.get_rng_state()[0],
                                fpemu_hip.next_offset(rmode))
        return super().apply(input, mode, inplace, scale, blocknorm, blocksize)

    @staticmethod
    def forward(ctx, input, mode='NONE', inplace=False, scale=1.0, blocknorm=False, blocksize=1):
        if mode == 'NONE':
            ctx.mark_dirty(input)
            return input
        if input.is_sparse:
            input = input.coalesce()
            values = input._values().contiguous()
            outputs = fpemu_hip.forward(values, mode, values.nelement(), inplace, scale, blocknorm, blocksize)
            if inplace:
                output = input
            else:
                output = torch.sparse_coo_tensor(input.indices(), outputs[0], input.size())
        else:
            dense = input.contiguous()
            outputs = fpemu_hip.forward(dense, mode, dense.nelement(), inplace, scale, blocknorm, blocksize)
            output = outputs[0]
        if inplace:
            ctx.mark_dirty(input)
        return output

    @staticmethod
    def backward(ctx, output_grad):
        # straight-through estimator
        return output_grad, None, None, None, None, None
