"""E5M2 emulator: every hooked tensor in E5M2 (reference mpemu/e5m2_emu.py).
Training recipe: weights / activations E5M2_RNE unscaled, all gradients E5M2_STOCHASTIC
(e5m2_emu.py:55-61); inference preset: weights per-channel-mean scaled (:407-414)."""
from .emulator_base import EmulatorBase


class E5M2Emulator(EmulatorBase):
    tag = "e5m2_emulator"
    hooks_layernorm = True
    defaults = {"emb": None, "wt": ("e5m2", "rne"), "iact": ("e5m2", "rne"), "oact": ("e5m2", "rne"),
                "igrad": ("e5m2", "stochastic"), "ograd": ("e5m2", "stochastic"),
                "wtgrad": ("e5m2", "stochastic")}
    calibration = {"emb": ("e5m2", "rne"), "wt": ("e5m2", "rne"), "iact": ("e5m2", "rne"), "oact": None}
    inference = {"emb": ("e5m2", "rne", "per-channel-mean"), "wt": ("e5m2", "rne", "per-channel-mean"),
                 "iact": ("e5m2", "rne"), "oact": None, "igrad": None, "ograd": None, "wtgrad": None}

    def __init__(self, model, optimizer, sparse_config=None, device="cuda", verbose=False, tensor_stats=False):
        super().__init__(model, optimizer, sparse_config, device=device, verbose=verbose,
                         tensor_stats=tensor_stats)
