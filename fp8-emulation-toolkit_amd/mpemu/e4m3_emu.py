"""E4M3 emulator (reference mpemu/e4m3_emu.py): inference-oriented; weights of Conv2d / Linear and
the inputs of every hooked module in E4M3_RNE; default inference preset = per-channel weights,
per-tensor input activations (e4m3_emu.py:181-185)."""
import torch

from .emulator_base import EmulatorBase


class E4M3Emulator(EmulatorBase):
    tag = "e4m3_emulator"
    training_recipe = False
    weights_only_for = (torch.nn.Conv2d, torch.nn.Linear)
    defaults = {"emb": ("e4m3", "rne"), "wt": ("e4m3", "rne"), "iact": ("e4m3", "rne"), "oact": None}
    calibration = {"emb": ("e4m3", "rne", "per-channel"), "wt": ("e4m3", "rne", "per-channel"),
                   "iact": ("e4m3", "rne", "per-tensor"), "oact": None}
    inference = dict(calibration)

    def __init__(self, model, optimizer, sparse_config=None, device="cuda", train=False, verbose=False,
                 tensor_stats=False):
        super().__init__(model, optimizer, sparse_config, device=device, train=train, verbose=verbose,
                         tensor_stats=tensor_stats)
