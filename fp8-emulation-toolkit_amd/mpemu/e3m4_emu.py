"""E3M4 emulator (reference mpemu/e3m4_emu.py): as the E4M3 emulator with the E3M4 format; its
construction-time default already scales weights per channel and inputs per tensor (:45-48)."""
import torch

from .emulator_base import EmulatorBase


class E3M4Emulator(EmulatorBase):
    tag = "e3m4_emulator"
    training_recipe = False
    weights_only_for = (torch.nn.Conv2d, torch.nn.Linear)
    defaults = {"emb": ("e3m4", "rne", "per-channel"), "wt": ("e3m4", "rne", "per-channel"),
                "iact": ("e3m4", "rne", "per-tensor"), "oact": None}
    calibration = {"emb": ("e3m4", "rne", "per-tensor"), "wt": ("e3m4", "rne", "per-tensor"),
                   "iact": ("e3m4", "rne", "per-tensor"), "oact": None}       # e3m4_emu.py:159-163
    inference = dict(defaults)

    def __init__(self, model, optimizer, sparse_config=None, device="cuda", verbose=False, tensor_stats=False):
        super().__init__(model, optimizer, sparse_config, device=device, train=False, verbose=verbose,
                         tensor_stats=tensor_stats)
