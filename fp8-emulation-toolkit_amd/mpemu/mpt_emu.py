"""mpemu.mpt_emu -- the user-facing facade (reference mpemu/mpt_emu.py):

    model, emulator = mpt_emu.initialize(model, optimizer, training_algo="hybrid" | "direct", ...)
    model, emulator = mpt_emu.quantize_model(model, dtype="e4m3" | "e5m2" | "e3m4" | "hybrid", ...)
    emulator.optimizer_step(optimizer)

Out of scope in this build and rejected loudly: sparsity / pruning (`pruning_algo` other than
"none", reference sparse_utils.py / adasparse.py).  Tensors must live on a ROCm device.
`hw_patch="simple"` is accepted on the GPU: the c-model GEMM / convolution of mpemu.cmodel.simple run
there (the reference only allows it with device='cpu', mpt_emu.py:150-151, because its c-model is CPU
code).
"""
from .e3m4_emu import E3M4Emulator
from .e4m3_emu import E4M3Emulator
from .e5m2_emu import E5M2Emulator
from .hybrid_emu import HybridEmulator


class MPTEmulator(object):
    valid_training_methods = ["direct", "hybrid"]
    valid_hw_patches = ["none", "simple"]
    valid_pruning_methods = ["none", "fine-grained", "unstructured", "adaptive", "auto"]

    def __init__(self, device='cuda', training_algo='direct', hw_patch='none', pruning_algo='none'):
        super(MPTEmulator, self).__init__()
        for what, value, valid in (("training method", training_algo, self.valid_training_methods),
                                   ("hardware emulation", hw_patch, self.valid_hw_patches),
                                   ("pruning method", pruning_algo, self.valid_pruning_methods)):
            if value.lower() not in valid:
                raise RuntimeError("mpt_emulator: the requested {} {} is not supported, use one of the "
                                   "following: {}".format(what, value, valid))
        if pruning_algo.lower() != "none":
            raise NotImplementedError("mpt_emulator (MI355X build): pruning / sparse training is out of scope")
        self.device = device
        self.training_algo = training_algo.lower()
        self.hw_patch = hw_patch.lower()
        self.pruning_method = pruning_algo.lower()
        self.wt_sparsity = 0.5
        self.grad_sparsity = 0.5
        self.emulator = None
        self.sparse_config = None

    # thin delegations, same names as the reference (mpt_emu.py:61-114)
    def blacklist_modules(self, list_modules):
        self.emulator.blacklist_modules(list_modules)

    def whitelist_modules(self, list_modules):
        self.emulator.whitelist_modules(list_modules)

    def optimizer_step(self, optimizer):
        self.emulator.optimizer_step(optimizer)

    def update_global_steps(self, global_steps):
        self.emulator.global_steps = global_steps

    def set_master_param_precision(self, master_params):
        self.emulator.set_master_param_precision(master_params)

    def set_embedding_precision(self, emb_precision, emb_norm=False):
        self.emulator.set_embedding_precision(emb_precision, emb_norm)

    def enable_tensor_stats(self, summary_writer=None):
        self.emulator.enable_tensor_stats(summary_writer)

    enable_tensorboard_logging = enable_tensor_stats

    def set_tensor_bindump_schedule(self, list_bindump_schedule):
        self.emulator.set_tensor_bindump_schedule(list_bindump_schedule)

    def set_target_sparsity_weight(self, wt_sparsity=0.5):
        print("mpt-emulator: set_target_sparsity_weight has no effect; sparse training is not enabled")

    def set_target_sparsity_gradient(self, grad_sparsity=0.5):
        print("mpt-emulator: set_target_sparsity_gradient has no effect; sparse training is not enabled")

    def fuse_bnlayers_and_quantize_model(self, model):
        self.emulator.is_training = False
        return self.emulator.fuse_layers_and_quantize_model(model)

    def set_calibration_qconfig(self):
        self.emulator.set_calibration_qconfig()

    def set_default_inference_qconfig(self):
        self.emulator.set_default_inference_qconfig()

    def __repr__(self):
        return repr(self.emulator)


def _check_device(device, hw_patch):
    if device != 'cuda' and not str(device).startswith("cuda"):
        raise RuntimeError("mpt_emulator (MI355X build): only ROCm devices are supported, got {}".format(device))


def initialize(model, optimizer, training_algo='direct', hw_patch='none', pruning_algo='none',
               list_exempt_layers=None, list_layers_output_fused=None, device="cuda", verbose=False):
    """Training entry point (reference mpt_emu.py:146-191)."""
    if model is None:
        raise RuntimeError("mpt_emulator: Undefined model and optimizer, call this after model and optimizer "
                           "are initilized.")
    _check_device(device, hw_patch)
    mpt = MPTEmulator(device=device, training_algo=training_algo, hw_patch=hw_patch, pruning_algo=pruning_algo)
    cls = HybridEmulator if mpt.training_algo == 'hybrid' else E5M2Emulator
    mpt.emulator = cls(model, optimizer, mpt.sparse_config, device=device, verbose=verbose)
    mpt.emulator.set_master_param_precision("fp16")
    mpt.emulator.enable_hw_patching(mpt.hw_patch.upper())
    mpt.emulator.prepare_model(model, list_exempt_layers, list_layers_output_fused)
    if verbose:
        mpt.emulator.print_config()
    return model, mpt


_INFERENCE_EMULATORS = {"E5M2": E5M2Emulator, "E4M3": E4M3Emulator, "E3M4": E3M4Emulator, "HYBRID": HybridEmulator}


def quantize_model(model, optimizer=None, dtype="none", calibrate=False, hw_patch="none", fuse_bn=False,
                   list_exempt_layers=None, list_layers_output_fused=None, device="cuda", verbose=False):
    """Post-training quantisation entry point (reference mpt_emu.py:193-231)."""
    if model is None:
        raise RuntimeError("mpt_emulator: Undefined model , call this after model is initilized.")
    _check_device(device, hw_patch)
    cls = _INFERENCE_EMULATORS.get(dtype.upper())
    if cls is None:
        # the reference leaves mpt.emulator = None here and dies on an AttributeError
        raise RuntimeError("mpt_emulator: the requested data type {} is not supported, use one of the "
                           "following: {}".format(dtype, sorted(_INFERENCE_EMULATORS)))
    mpt = MPTEmulator(device=device, hw_patch=hw_patch)
    mpt.emulator = cls(model, optimizer, None, device=device, verbose=verbose)
    mpt.emulator.is_training = False
    if fuse_bn:
        # the reference calls this before any emulator exists (mpt_emu.py:203-204) and fails
        from .scale_shift import replace_batchnorms_with_scaleshifts
        model = replace_batchnorms_with_scaleshifts(model).to(device)
        mpt.emulator.model = model
    if calibrate:
        mpt.set_calibration_qconfig()
    else:
        mpt.set_default_inference_qconfig()
    mpt.emulator.enable_hw_patching(mpt.hw_patch.upper())
    mpt.emulator.prepare_model(model, list_exempt_layers, list_layers_output_fused)
    if verbose:
        mpt.emulator.print_config()
    return model, mpt
