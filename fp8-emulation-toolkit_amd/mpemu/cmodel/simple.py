"""mpemu.cmodel.simple -- the "simple" hardware c-model GEMM ops, MI355X build.

Mirror of the reference's mpemu/cmodel/simple.py for the matrix products: `addmm`, `matmul`, `mm`
with the same signatures and fallbacks, backed by the gfx950 kernels of csrc/fpemu_gemm.hip instead of
the CPU extension simple_gemm_dev.  Every output element is a k-ascending fp32 accumulation chain
(no FP8 rounding inside: operands are pre-quantised by the hooks).

Accumulation flavour.  Built with its own flags on an FMA-capable x86 host the reference contracts the
engine's add(c, mul(a, b)) into vfmadd231ps, i.e. an fmaf chain; on a host without FMA the same source
rounds the product and the sum separately (oracle/build_ref.py builds both; tests pin both).
`set_accumulation("fma")` (default, = the reference as built here; runs on the fp32 MFMA, which is
bit-for-bit that chain) or `set_accumulation("separate")` (explicit per-step rounding, VALU kernel).

`conv2d` is the c-model's direct convolution (reference simple_conv2d_dev conv2d_fp / _bp / _wu): square
stride and padding, no dilation, no groups, as the reference.
Divergences: backward zero-initialises the gradient buffers (the reference accumulates into
torch.empty_like, simple.py:163-164); unsupported conv arguments raise ValueError instead of sys.exit;
the batch is convolved image by image also when C or K < 64 (the reference's packing collapses it).

Output maxima.  With `collect_output_amax(True)` (the module hooks switch it on around a hooked module whose output
is quantised per tensor) `mm` / `matmul` / `conv2d` run the kernels' `_amax` forms: the epilogue folds max |output|
into a device word and leaves it with the binding (`_binding.note_amax`), so the quantisation of that output
(quantize_module_output, reference qutils.py:607-609) is one 8-byte-per-element cast instead of an amax pass plus
the cast.  Same bits either way.  `addmm` adds beta * input after the product, so its maximum is not known.
"""
import warnings

import torch
from torch.autograd import Function

from ..pytquant.hip import _binding as _hip

fallback_addmm = torch.addmm
fallback_matmul = torch.matmul
fallback_mm = torch.mm

_variant = _hip.GEMM_FMA
_collect = False


def collect_output_amax(on):
    """Switch the epilogue amax of mm / matmul / conv2d on or off; returns the previous setting."""
    global _collect
    prev, _collect = _collect, bool(on)
    return prev


def _amax_word(like):
    return torch.zeros(1, dtype=torch.int32, device=like.device) if _collect else None


def set_accumulation(kind):
    """'fma': fmaf chain (reference on FMA hosts; MFMA kernel).  'separate': multiply and add rounded
    separately (reference on hosts without FMA; VALU kernel)."""
    global _variant
    _variant = {"fma": _hip.GEMM_FMA, "separate": _hip.GEMM_SEPARATE}[kind]


def get_accumulation():
    return "fma" if _variant == _hip.GEMM_FMA else "separate"


def is_transposed(input):
    """(matrix to hand to the GEMM, transposed flag): a transposed view of a contiguous matrix is passed as
    a flag, anything else is made contiguous (reference simple.py:21-27)."""
    if input.is_contiguous():
        return input, False
    if input.t().is_contiguous():
        return input.t(), True
    return input.contiguous(), False


def _gemm(out, a, b, alpha, a_trans, b_trans, amax=None):
    return _hip.gemm_(out, a, b, alpha, a_trans, b_trans, _variant, amax=amax)


def _on_gpu_f32(*ts):
    return all(t.dtype == torch.float32 and t.is_cuda for t in ts)


def _mat2(a, b):
    """Both 2-D with matching inner dimension."""
    return a.dim() == 2 and b.dim() == 2 and a.size(1) == b.size(0)


def _fresh(m, n, like):
    return torch.zeros([m, n], device=like.device)


def _decline(op, fallback, shapes, *args, **kwargs):
    warnings.warn("simple.%s: unsupported operands %s -- using torch.%s" % (op, shapes, op))
    return fallback(*args, **kwargs)


def addmm(input, mat1, mat2, beta=1.0, alpha=1.0, out=None):
    if not (_on_gpu_f32(input, mat1, mat2) and _mat2(mat1, mat2)):
        return _decline("addmm", fallback_addmm, (tuple(input.size()), tuple(mat1.size()), tuple(mat2.size())),
                        input, mat1, mat2, beta=beta, alpha=alpha, out=out)
    a_mat, a_trans = is_transposed(mat1)
    b_mat, b_trans = is_transposed(mat2)
    output = out if out is not None else _fresh(mat1.size(0), mat2.size(1), mat1)
    return SimpleAddmm.apply(output, input, a_mat, b_mat, alpha, beta, a_trans, b_trans)


def mm(input, mat2, out=None):
    if not (_on_gpu_f32(input, mat2) and _mat2(input, mat2)):
        return _decline("mm", fallback_mm, (tuple(input.size()), tuple(mat2.size())), input, mat2, out=out)
    a_mat, a_trans = is_transposed(input)
    b_mat, b_trans = is_transposed(mat2)
    fresh = out is None
    output = _fresh(input.size(0), mat2.size(1), input) if fresh else out
    word = _amax_word(output) if fresh and output.is_contiguous() else None   # (a given `out` may hold other values)
    res = SimpleMatmul.apply(output, a_mat, b_mat, 1.0, a_trans, b_trans, word)
    if word is not None:
        _hip.note_amax(res, word)
    return res


def matmul(input, other, out=None):
    if _on_gpu_f32(input, other) and _mat2(input, other):
        return mm(input, other, out=out)
    if _on_gpu_f32(input, other) and input.dim() == 3 and other.dim() == 2 and input.size(2) == other.size(0):
        # a stack of matrices times one matrix: one c-model GEMM per slice (reference simple.py:52-63)
        b_mat, b_trans = is_transposed(other)
        outs = out if out is not None else torch.zeros([input.size(0), input.size(1), other.size(1)],
                                                       device=input.device)
        word = _amax_word(outs) if out is None else None   # every slice folds its maximum into the one word
        res = []
        for i in range(input.size(0)):
            a_mat, a_trans = is_transposed(input[i])
            res.append(SimpleMatmul.apply(outs[i], a_mat, b_mat, 1.0, a_trans, b_trans, word))
        stacked = torch.stack(res)
        if word is not None:
            _hip.note_amax(stacked, word)   # a copy of the slices: the same values, the same maximum
        return stacked
    return _decline("matmul", fallback_matmul, (tuple(input.size()), tuple(other.size())), input, other, out=out)


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


def conv2d(input, weight, bias=None, stride=1, padding=0, dilation=1, groups=1):
    stride, padding, dilation = _pair(stride), _pair(padding), _pair(dilation)
    if dilation[0] > 1 or dilation[1] > 1:
        raise ValueError("simple_conv2d does not support dilated convolutions.")
    if padding[0] != padding[1]:
        raise ValueError("simple_conv2d does not support non-uniform padding; pad_h must be equal to pad_w.")
    if stride[0] != stride[1]:
        raise ValueError("simple_conv2d applies stride[0] in both dimensions; non-square strides are not supported.")
    if groups > 1:
        raise ValueError("simple_conv2d does not support grouped convolutions; set groups to 1.")
    N, C, H, W = input.size()
    K, _, R, S = weight.size()
    h_out = (H + 2 * padding[0] - dilation[0] * (R - 1) - 1) // stride[0] + 1
    w_out = (W + 2 * padding[1] - dilation[1] * (S - 1) - 1) // stride[1] + 1
    output = torch.empty([N, K, h_out, w_out], device=input.device)
    word = _amax_word(output)
    res = SimpleConv2dFunction.apply(output, input.contiguous(), weight.contiguous(), bias, stride, padding,
                                     dilation, groups, word)
    if word is not None:
        _hip.note_amax(res, word)
    return res


def _grads(grad_output, mat1, mat2, alpha, a_trans, b_trans, need1=True, need2=True):
    """d/d(mat1), d/d(mat2) of alpha * op(mat1) @ op(mat2), each one cmodel GEMM (simple.py:155-178).  A gradient
    autograd does not ask for (ctx.needs_input_grad: a frozen weight, an input that requires none) is not computed --
    the reference computes both regardless and autograd drops the unused one."""
    g1 = g2 = None
    go, o_trans = is_transposed(grad_output)
    if need1:
        g1 = torch.zeros_like(mat1)
        if a_trans:
            _gemm(g1, mat2, go, alpha, b_trans, not o_trans)
        else:
            _gemm(g1, go, mat2, alpha, o_trans, not b_trans)
    if need2:
        g2 = torch.zeros_like(mat2)
        if b_trans:
            _gemm(g2, go, mat1, alpha, not o_trans, a_trans)
        else:
            _gemm(g2, mat1, go, alpha, not a_trans, o_trans)
    return g1, g2


class SimpleAddmm(Function):
    @staticmethod
    def forward(ctx, output, input, mat1, mat2, alpha, beta, a_trans, b_trans):
        ctx.save_for_backward(mat1, mat2)
        ctx.a_trans, ctx.b_trans, ctx.alpha = a_trans, b_trans, alpha
        _gemm(output, mat1, mat2, alpha, a_trans, b_trans)
        output += beta * input
        ctx.mark_dirty(output)
        return output

    @staticmethod
    def backward(ctx, grad_output):
        mat1, mat2 = ctx.saved_tensors
        g1, g2 = _grads(grad_output, mat1, mat2, ctx.alpha, ctx.a_trans, ctx.b_trans, ctx.needs_input_grad[2],
                        ctx.needs_input_grad[3])
        return grad_output, grad_output, g1, g2, None, None, None, None


class SimpleMatmul(Function):
    @staticmethod
    def forward(ctx, output, mat1, mat2, alpha, a_trans, b_trans, amax=None):
        ctx.save_for_backward(mat1, mat2)
        ctx.a_trans, ctx.b_trans, ctx.alpha = a_trans, b_trans, alpha
        _gemm(output, mat1, mat2, alpha, a_trans, b_trans, amax)
        ctx.mark_dirty(output)
        return output

    @staticmethod
    def backward(ctx, grad_output):
        mat1, mat2 = ctx.saved_tensors
        g1, g2 = _grads(grad_output, mat1, mat2, ctx.alpha, ctx.a_trans, ctx.b_trans, ctx.needs_input_grad[1],
                        ctx.needs_input_grad[2])
        return grad_output, g1, g2, None, None, None, None


class SimpleConv2dFunction(Function):
    @staticmethod
    def forward(ctx, output, inputs, weights, bias, stride, padding, dilation, groups, amax=None):
        ctx.save_for_backward(inputs, weights)
        ctx.stride, ctx.padding = stride, padding
        _hip.conv2d_fp_(output, inputs, weights, None if bias is None else bias.contiguous(), stride[0], padding[0],
                        _variant, amax=amax)
        ctx.mark_dirty(output)
        return output

    @staticmethod
    def backward(ctx, grad_output):
        inputs, weights = ctx.saved_tensors
        grad_output = grad_output.contiguous()
        grad_inp = grad_wts = None
        # a gradient autograd does not ask for is not computed (a network's first convolution: the images need none --
        # 0.36 ms of backward-data at batch 32; the reference computes both regardless and autograd drops the unused one)
        if ctx.needs_input_grad[1]:
            grad_inp = torch.empty_like(inputs)
            _hip.conv2d_bp_(grad_inp, grad_output, weights, ctx.stride[0], ctx.padding[0], _variant)
        if ctx.needs_input_grad[2]:
            grad_wts = torch.empty_like(weights)
            _hip.conv2d_wu_(grad_wts, grad_output, inputs, ctx.stride[0], ctx.padding[0], _variant)
        return grad_output, grad_inp, grad_wts, None, None, None, None, None, None
