"""Hybrid emulator (reference mpemu/hybrid_emu.py): E4M3 forward, E5M2 backward.
Training recipe: weights / input / output activations E4M3_RNE with per-tensor max scaling,
input / output / weight gradients E5M2_STOCHASTIC unscaled (hybrid_emu.py:56-61); inference
preset: E3M4 per-channel weights, E4M3 per-tensor inputs (:406-413)."""
from .emulator_base import EmulatorBase


class HybridEmulator(EmulatorBase):
    tag = "hybrid_emulator"
    hooks_layernorm = True
    defaults = {"emb": None, "wt": ("e4m3", "rne", "per-tensor"), "iact": ("e4m3", "rne", "per-tensor"),
                "oact": ("e4m3", "rne", "per-tensor"), "igrad": ("e5m2", "stochastic"),
                "ograd": ("e5m2", "stochastic"), "wtgrad": ("e5m2", "stochastic")}
    calibration = {"emb": ("e3m4", "rne", "per-channel"), "wt": ("e3m4", "rne", "per-channel"),
                   "iact": ("e4m3", "rne", "per-tensor"), "oact": None}
    inference = dict(calibration, igrad=None, ograd=None, wtgrad=None)

    def __init__(self, model, optimizer, sparse_config=None, device="cuda", verbose=False, tensor_stats=False):
        super().__init__(model, optimizer, sparse_config, device=device, verbose=verbose,
                         tensor_stats=tensor_stats)
