"""Hook targets for functional ops.  Same class names, constructor arguments and forward
signatures as the reference (module_wrappers/matmul.py:12-32, eltwise.py:12-61,
aggregate.py:12-35).  `EltwiseMul` constructs here (the reference's __init__ has a NameError,
eltwise.py:34)."""
import torch
import torch.nn as nn


class Matmul(nn.Module):
    def forward(self, a: torch.Tensor, b: torch.Tensor):
        return torch.matmul(a, b)


class BatchMatmul(nn.Module):
    def forward(self, a: torch.Tensor, b: torch.Tensor):
        return torch.bmm(a, b)


class AddMatmul(nn.Module):
    def forward(self, input: torch.Tensor, mat1: torch.Tensor, mat2: torch.Tensor, beta=1, alpha=1):
        return torch.addmm(input, mat1, mat2, beta=beta, alpha=alpha)


class _Eltwise(nn.Module):
    def __init__(self, inplace=False):
        super().__init__()
        self.inplace = inplace

    def extra_repr(self) -> str:
        return 'inplace={}'.format(self.inplace)


class EltwiseAdd(_Eltwise):
    def forward(self, *input):
        res = input[0]
        for t in input[1:]:
            if self.inplace:
                res += t
            else:
                res = res + t
        return res


class EltwiseMul(_Eltwise):
    def forward(self, *input):
        res = input[0]
        for t in input[1:]:
            if self.inplace:
                res *= t
            else:
                res = res * t
        return res


class EltwiseDiv(_Eltwise):
    def forward(self, x: torch.Tensor, y):
        return x.div_(y) if self.inplace else x.div(y)


class Norm(nn.Module):
    def __init__(self, p='fro', dim=None, keepdim=False):
        super().__init__()
        self.p, self.dim, self.keepdim = p, dim, keepdim

    def forward(self, x: torch.Tensor):
        return torch.norm(x, p=self.p, dim=self.dim, keepdim=self.keepdim)

    def extra_repr(self) -> str:
        return 'p={}, dim={}, keepdim: {}'.format(self.p, self.dim, self.keepdim)


class Mean(nn.Module):
    def __init__(self, *args, **kwargs):
        super().__init__()
        self.args, self.kwargs = args, kwargs

    def forward(self, x: torch.Tensor):
        return torch.mean(x, *self.args, **self.kwargs)

    def extra_repr(self) -> str:
        return 'args={}, kwargs={}'.format(self.args, self.kwargs)


class SparseLinear(nn.Linear):
    """Type marker kept for the emulators' whitelists.  The reference's adaptive-sparsity training
    (module_wrappers/adasparse.py) is a sparsity research feature outside this build's scope
    (SURVEY section 2, row 11): this class is a dense Linear."""


class SparseConv2d(nn.Conv2d):
    """See SparseLinear: dense Conv2d, kept as a hook-target type."""
