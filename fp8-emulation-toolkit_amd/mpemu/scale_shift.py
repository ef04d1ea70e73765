"""mpemu.scale_shift -- per-channel affine layer that replaces an inference-mode BatchNorm2d
(reference mpemu/scale_shift.py:12-73).  Pure torch; no kernel of the cast path lives here.

    y[:, c] = x[:, c] * weight[c] + bias[c],
    weight = gamma / sqrt(running_var + eps),  bias = beta - gamma * running_mean / sqrt(running_var + eps)
"""
import torch


class ScaleShift(torch.nn.Module):
    def __init__(self, num_features):
        super().__init__()
        self.num_features = num_features
        self.weight = torch.nn.Parameter(torch.ones(num_features))
        self.bias = torch.nn.Parameter(torch.zeros(num_features))

    def reset_parameters(self):
        torch.nn.init.ones_(self.weight)
        torch.nn.init.zeros_(self.bias)

    def forward(self, input):
        # channel is dim 1 (dim 0 is the batch): move it last so weight/bias broadcast, then back.
        # Same arithmetic order as the reference (multiply, then add, on the transposed view).
        moved = input.transpose(1, -1)
        return (moved * self.weight + self.bias).transpose(1, -1)

    def __repr__(self):
        return 'ScaleShift({})'.format(self.num_features)

    @staticmethod
    def generate_from_batchnorm(module: torch.nn.BatchNorm2d):
        ss = ScaleShift(module.num_features)
        with torch.no_grad():
            denom = torch.sqrt(module.running_var + module.eps)
            ss.weight.copy_(module.weight.div(denom))
            ss.bias.copy_(module.bias - module.weight.mul(module.running_mean).div(denom))
        return ss


def replace_batchnorms_with_scaleshifts(model):
    """Recursively swap every BatchNorm2d for its ScaleShift equivalent; returns the model."""
    if isinstance(model, torch.nn.BatchNorm2d):
        return ScaleShift.generate_from_batchnorm(model)
    for name, child in model.named_children():
        setattr(model, name, replace_batchnorms_with_scaleshifts(child))
    return model
