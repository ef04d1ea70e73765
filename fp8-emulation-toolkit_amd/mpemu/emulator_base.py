"""Shared machinery of the per-recipe emulators (E5M2 / E4M3 / E3M4 / Hybrid).

The reference carries four near-identical classes (mpemu/e5m2_emu.py, e4m3_emu.py, e3m4_emu.py,
hybrid_emu.py); here one base class holds the behaviour and each recipe is a small table:
which module types are hooked, the default TensorQuantConfig per tensor role, and the
calibration / inference presets.  The emulators never touch numbers themselves: they decide which
tensors get which mode string, install the qutils hooks, and re-quantise master parameters after
an optimizer step (all casts run in the HIP kernels through FPEmuOp / quantize_tensor).

Divergences from the reference (deliberate):
  * apex is optional.  With an apex-stashed optimizer the master copies are `amp.master_params`;
    with a plain torch optimizer they are the optimizer's own parameters (the reference raises
    RuntimeError unless the optimizer went through amp.initialize, hybrid_emu.py:69-79);
  * blacklist_modules / whitelist_modules pass the model to create_or_update_hooks (the reference
    omits the argument and raises TypeError, e5m2_emu.py:89);
  * tensorboard statistics and binary tensor dumps are accepted as settings and ignored.
`emb_norm` keeps the reference's call exactly, including its positional slip: FPEmuOp.apply(param, mode,
True, emb_norm, cols) binds scale=emb_norm and blocknorm=cols, blocksize=1 (hybrid_emu.py:119), i.e. block
norm with one element per block -- that is what the reference executes, so that is what runs here.
"""
import torch

from .module_wrappers import (AddMatmul, BatchMatmul, EltwiseAdd, EltwiseDiv, EltwiseMul, Matmul,
                              SparseConv2d, SparseLinear)
from .qutils import (ModuleQuantConfig, TensorQuantConfig, add_quantization_hooks,
                     get_or_update_model_quant_config_dict, quantize_model_weights,
                     reset_quantization_setup, set_quantize_weights_flag)
from .scale_shift import replace_batchnorms_with_scaleshifts

_COMMON_TYPES = [torch.nn.Conv2d, torch.nn.Linear, torch.nn.Embedding, torch.nn.EmbeddingBag]
_WRAPPER_TYPES = [Matmul, BatchMatmul, AddMatmul, EltwiseAdd, EltwiseMul, EltwiseDiv, SparseConv2d, SparseLinear]
_EMB_MODES = {"e5m2": "E5M2_STOCHASTIC", "BF8": "E5M2_STOCHASTIC",
              "e4m3": "E4M3_STOCHASTIC", "HF8": "E4M3_STOCHASTIC",
              "e4m3_2": "E3M4_STOCHASTIC", "HF8_2": "E3M4_STOCHASTIC"}


def _cfg(spec):
    """spec: None | (dtype, scheme) | (dtype, scheme, scaling)."""
    return None if spec is None else TensorQuantConfig(*spec)


class EmulatorBase(object):
    # ---- recipe tables, overridden by the subclasses ---------------------------------------
    tag = "emulator"
    hooks_layernorm = False          # LayerNorm in the whitelist (e5m2 / hybrid recipes)
    weights_only_for = None          # None: any module with a .weight keeps wt_qconfig;
    #                                  else the exact types that do (e4m3 / e3m4 recipes)
    training_recipe = True           # False: constructor's is_training does not follow model.training
    defaults = {}                    # role -> spec at construction
    calibration = {}                 # roles overwritten by set_calibration_qconfig
    inference = {}                   # roles overwritten by set_default_inference_qconfig
    _ROLES = ("emb", "wt", "iact", "oact", "igrad", "ograd", "wtgrad")

    def __init__(self, model, optimizer, sparse_config=None, device="cuda", train=None, verbose=False,
                 tensor_stats=False):
        super(EmulatorBase, self).__init__()
        self.whitelist = list(_COMMON_TYPES)
        if self.hooks_layernorm:
            self.whitelist.append(torch.nn.LayerNorm)
        self.whitelist += _WRAPPER_TYPES
        self.blacklist = []
        self.list_unpatched = []
        self.list_exempt_layers = None
        self.list_layers_output_fused = None
        self.using_apex = optimizer is not None and hasattr(optimizer, "_amp_stash")
        self.use_fp16_master = False
        self.use_e5m2_master = False
        self.emb_master_mode = None
        self.emb_norm = False
        self.device = device
        self.model = model
        self.global_steps = 0
        if train is not None:
            self.is_training = bool(train)
        else:
            self.is_training = bool(model.training) if self.training_recipe else False
        self.patch_ops = False
        self.patch_impl = "NONE"
        self.patchlist = ["SIMPLE"]
        self.sparse_config = sparse_config
        self.data_emulation = True
        self.mod_qconfig = None
        self.model_qconfig_dict = None
        for role in self._ROLES:
            setattr(self, role + "_qconfig", _cfg(self.defaults.get(role)))
        self.hook_handles = None
        self.bin_dump_enable = None
        self.bin_dump_disable = None
        self.verbose = verbose
        self.tensor_stats = tensor_stats
        self.tb_writer = None

    # ---- module-type lists -------------------------------------------------------------------
    def _move(self, list_modules, src, dst, what):
        if self.verbose:
            print("{} modules {}".format(what, list_modules))
        for m in list_modules:
            dst.append(m)
            src.remove(m)
        if self.model is not None and self.mod_qconfig is not None:
            self.create_or_update_hooks(self.model)
        if self.verbose:
            self.print_config()

    def blacklist_modules(self, list_modules):
        self._move(list_modules, self.whitelist, self.blacklist, "Blacklist-ing")

    def whitelist_modules(self, list_modules):
        self._move(list_modules, self.blacklist, self.whitelist, "Whitelist-ing")

    # ---- master parameters ---------------------------------------------------------------------
    def _master_params(self, optimizer):
        if self.using_apex and hasattr(optimizer, "_master_params_to_model_params"):
            from apex import amp
            return list(amp.master_params(optimizer))
        return [p for g in optimizer.param_groups for p in g["params"]]

    @staticmethod
    def _cast_run(params, mode):
        """`param.data = FPEmuOp.apply(param.data, mode, True)` for every parameter of the run, batched: runs of
        parameters that share a dtype and device go through the multi-tensor cast (same bits, same call
        offsets as the loop; 32 tensors per launch instead of one)."""
        from .pytquant import fpemu_cuda
        from .pytquant.hip import _binding as hip
        fmt, rmode = hip.parse_mode(mode)
        i = 0
        while i < len(params):
            j = i + 1
            head = params[i].data
            while j < len(params) and params[j].data.dtype == head.dtype and params[j].data.device == head.device:
                j += 1
            run = [p.data for p in params[i:j]]
            if j - i > 1 and head.is_cuda and all(t.is_contiguous() for t in run):
                hip.cast_multi_(run, fmt, rmode)
            else:
                for p in params[i:j]:
                    p.data = fpemu_cuda.FPEmuOp.apply(p.data, mode, True)
            i = j

    def update_fp16_master_params(self, optimizer):
        """FLOAT16_STOCHASTIC on every master parameter, plus the embedding format on parameters
        whose name contains "embeddings" (hybrid_emu.py:110-176).  Same sequence of casts and call offsets
        as the reference's loop; the FLOAT16 casts between two embedding parameters are batched."""
        from .pytquant import fpemu_cuda
        names = [n for n, _ in self.model.named_parameters()]
        pending = []
        for param, name in zip(self._master_params(optimizer), names):
            if self.emb_master_mode is None or "embeddings" not in name:
                pending.append(param)
                continue
            self._cast_run(pending, "FLOAT16_STOCHASTIC")
            pending = []
            param.data = fpemu_cuda.FPEmuOp.apply(param.data, "FLOAT16_STOCHASTIC", True)
            if self.emb_norm:
                # (1-D "embeddings" parameters such as LayerNorm weights: the reference's
                # param.size()[1] raises IndexError; only the value's truth is used, so pass 1)
                cols = param.size()[1] if param.dim() > 1 else 1
                param.data = fpemu_cuda.FPEmuOp.apply(param.data, self.emb_master_mode, True, self.emb_norm, cols)
            else:
                param.data = fpemu_cuda.FPEmuOp.apply(param.data, self.emb_master_mode, True)
        self._cast_run(pending, "FLOAT16_STOCHASTIC")

    def update_e5m2_master_params(self, optimizer):
        """FLOAT16_STOCHASTIC then E5M2_STOCHASTIC on every master parameter (hybrid_emu.py:178-190)."""
        from .pytquant import fpemu_cuda
        for param in self._master_params(optimizer):
            param.data = fpemu_cuda.FPEmuOp.apply(param.data, "FLOAT16_STOCHASTIC", True)
            param.data = fpemu_cuda.FPEmuOp.apply(param.data, "E5M2_STOCHASTIC", True)

    def update_master_params(self, optimizer):
        if self.use_fp16_master:
            self.update_fp16_master_params(optimizer)
        elif self.use_e5m2_master:
            self.update_e5m2_master_params(optimizer)

    def optimizer_step(self, optimizer):
        optimizer.step()
        self.update_master_params(optimizer)
        self.global_steps += 1

    def set_master_param_precision(self, master_params):
        if master_params in ("fp16", "FP16"):
            self.use_fp16_master, self.use_e5m2_master = True, False
        if master_params in ("e5m2", "BF8"):
            self.use_fp16_master, self.use_e5m2_master = False, True
        if self.verbose:
            print("{}: master parameters fp16={} e5m2={}".format(self.tag, self.use_fp16_master, self.use_e5m2_master))

    def set_embedding_precision(self, emb_precision, emb_norm=False):
        if emb_precision in _EMB_MODES:
            self.emb_master_mode = _EMB_MODES[emb_precision]
            if emb_norm:
                self.emb_norm = emb_norm

    # kept as attributes the reference exposes
    use_e5m2_emb = property(lambda self: self.emb_master_mode == "E5M2_STOCHASTIC")
    use_e4m3_emb = property(lambda self: self.emb_master_mode == "E4M3_STOCHASTIC")
    use_e3m4_emb = property(lambda self: self.emb_master_mode == "E3M4_STOCHASTIC")

    # ---- hooks -----------------------------------------------------------------------------------
    def create_or_update_hooks(self, model):
        if self.hook_handles:
            for h in self.hook_handles:
                h.remove()
        d = get_or_update_model_quant_config_dict(model, self.whitelist, self.mod_qconfig,
                                                  model_qconfig_dict=self.model_qconfig_dict, override=True)
        self.model_qconfig_dict = d
        exempt = self.list_exempt_layers or []
        for name in exempt:
            d.pop(name, None)
        fused = self.list_layers_output_fused or []
        for name, module in model.named_modules():
            if name not in d:
                continue
            kind = type(module)
            if name in fused and name not in exempt:
                d[name].oact_qconfig = None
                d[name].ograd_qconfig = None
            if kind is torch.nn.Conv2d and (module.in_channels < 64 or module.out_channels < 64):
                d[name].patch_ops = False
                d[name].patch_impl = "NONE"
                self.list_unpatched += [name]
            keeps_weight = (hasattr(module, 'weight') if self.weights_only_for is None
                            else kind in self.weights_only_for)
            if not keeps_weight:
                d[name].wt_qconfig = None
                d[name].wtgrad_qconfig = None
            if kind in (torch.nn.Embedding, torch.nn.EmbeddingBag):
                d[name].wt_qconfig = self.emb_qconfig
                d[name].iact_qconfig = None
                d[name].igrad_qconfig = None
                d[name].ograd_qconfig = None
                d[name].oact_qconfig = None
            if kind is torch.nn.LayerNorm:
                d[name].wt_qconfig = None
            if kind is BatchMatmul:
                d[name].wt_qconfig = None
                d[name].wtgrad_qconfig = None
                d[name].oact_qconfig = None
                d[name].ograd_qconfig = None

        reset_quantization_setup(model, d)
        self.hook_handles = add_quantization_hooks(model, d, is_training=self.is_training)
        if not self.is_training:
            if self.verbose:
                print("{}: quantizing model weights..".format(self.tag))
            quantize_model_weights(model, d)
            set_quantize_weights_flag(model, d, False)

    def prepare_model(self, model, list_exempt_layers, list_layers_output_fused):
        mq = ModuleQuantConfig(wt_qconfig=self.wt_qconfig, iact_qconfig=self.iact_qconfig,
                               oact_qconfig=self.oact_qconfig, igrad_qconfig=self.igrad_qconfig,
                               ograd_qconfig=self.ograd_qconfig, wtgrad_qconfig=self.wtgrad_qconfig,
                               patch_ops=self.patch_ops)
        mq.device = self.device
        mq.patch_impl = self.patch_impl
        mq.sparse_config = self.sparse_config
        self.mod_qconfig = mq
        self.model = model
        self.list_exempt_layers = list_exempt_layers
        self.list_layers_output_fused = list_layers_output_fused
        self.create_or_update_hooks(model)

    # ---- presets -----------------------------------------------------------------------------------
    def _apply(self, table):
        for role, spec in table.items():
            setattr(self, role + "_qconfig", _cfg(spec))

    def set_calibration_qconfig(self):
        self._apply(self.calibration)

    def set_default_inference_qconfig(self):
        self._apply(self.inference)

    # ---- settings accepted for compatibility -------------------------------------------------------
    def enable_tensor_stats(self, summary_writer=None):
        self.tb_writer = summary_writer
        self.tensor_stats = False  # tensorboard statistics are outside this build's scope

    def set_tensor_bindump_schedule(self, list_bindump_schedule):
        if list_bindump_schedule is not None:
            self.bin_dump_enable = list(list_bindump_schedule)
            self.bin_dump_disable = [s + 1 for s in list_bindump_schedule]

    def enable_hw_patching(self, patch_ops):
        if patch_ops == 'NONE':
            return
        if patch_ops not in self.patchlist:
            raise RuntimeError("{}: HW patching is not supported for {}, supported list of options : {}".format(
                self.tag, patch_ops, self.patchlist))
        self.patch_ops = True
        self.patch_impl = patch_ops

    def fuse_batchnorm_with_convolution(self, model):
        from torch.nn.utils.fusion import fuse_conv_bn_eval
        seen = []
        for name, module in model.named_children():
            if list(module.named_children()):
                self.fuse_batchnorm_with_convolution(module)
            if isinstance(module, torch.nn.BatchNorm2d):
                if seen and isinstance(seen[-1][1], torch.nn.Conv2d):
                    setattr(model, seen[-1][0], fuse_conv_bn_eval(seen[-1][1], module))
                    setattr(model, name, torch.nn.Identity())
            else:
                seen.append((name, module))
        return model

    def fuse_layers_and_quantize_model(self, model):
        if self.is_training:
            print("Warning : emulator.is_training is set to True, returning the model unchanged")
            return model
        model = replace_batchnorms_with_scaleshifts(model)
        self.is_training = False
        self.set_default_inference_qconfig()
        self.prepare_model(model, self.list_exempt_layers, self.list_layers_output_fused)
        model = model.to(self.device)
        if self.verbose:
            self.print_config()
        return model

    def print_config(self):
        for key in self.model_qconfig_dict:
            print("{} {:40s}".format(self.model_qconfig_dict[key], key))

    def __repr__(self):
        return "[Configured to run {} on {}, using AMP: {}]".format(
            "training" if self.is_training else "inference", self.device, str(self.using_apex))
