"""mpemu.qutils -- tensor/module quantisation configs, the scale policy and the module hooks.

Host-side mirror of the reference's mpemu/qutils.py for the fpemu cast path, MI355X build.
Same names, arguments and results:

  TensorQuantConfig / ModuleQuantConfig / ModuleQuantParams   qutils.py:22-214
  fpemu_device_fn                                             qutils.py:249-261
  quantize_tensor                                             qutils.py:264-345
  model-config / hook helpers                                 qutils.py:351-365, 458-475, 478-660

What differs underneath: every cast runs in the gfx950 HIP kernels (pytquant/hip); the
per-tensor "max" policy computes amax, the scale and the cast on the device in one enqueue
(no `if scale > ...` host sync, qutils.py:285); per-channel "max" does the same per channel.
Results are bit-identical to the reference's CPU path (tests/golden/qtensor.npz).

With `set_amax_process_group(group)` a per-tensor "max" cast treats the tensor as one shard
of a larger tensor: the local amax word is all-reduced (MAX) over the group before the cast,
so every rank derives the same scale, and the stochastic noise is indexed by the shard's GLOBAL
position (`elem_base`), so the sharded result is the single-device result bit for bit
(DESIGN.md "Multi-GPU").  Per-channel / fine-grained "max" scaling under a group needs to know how
the rows lie across the ranks (`layout`): "channels" -- dim 0 is sharded, every rank holds whole
rows, no exchange; "flat" -- the tensor is this rank's contiguous piece of the flattened logical
[rows, cols] tensor, rows may straddle ranks: one all-reduce(MAX) of the vector of row maxima.
`quantize_model_weights` under a group issues ONE collective for all per-tensor weights.
"""
from collections import OrderedDict
import copy
import os
import threading

import torch

from .pytquant.hip import _binding as _hip

_FLT_MAX = {"e5m2": 57344.0, "e4m3": 448.0, "e4m3_ieee": 240.0, "e3m4": 30.0, "fp4": 1.0}
_FLT_MIN = {"e5m2": 1.5258789E-05, "e4m3": 1.9531250E-03, "e3m4": 1.5625000E-02, "fp4": 0.000244140625}

_amax_group = None
_amax_elem_base = None
_amax_layout = None
_amax_logical = None
_amax_row_base = None
_NO_SCALE_FMTS = (_hip.FMT_BF16, _hip.FMT_FP16, _hip.FMT_FP16_DAZ)  # their reference kernels take no scale


def set_amax_process_group(group, elem_base=None, layout=None, logical=None, row_base=None):
    """group: a torch.distributed process group (or None to go back to single-device scaling).

    elem_base: where this rank's shard starts in the logical tensor, for the stochastic noise index (and, with
    layout "flat", for the row it starts in) -- None: equal-sized contiguous shards, `rank * tensor.numel()`; an int;
    or a callable `elem_base(tensor) -> int` for ragged layouts (mpemu.sharded.shard_bounds gives the starts).

    layout: how per-channel / fine-grained tensors lie across the ranks (per-tensor scaling does not need it) --
      None       : such calls raise under a group (a silent local row maximum would be wrong for straddling rows);
      "channels" : dim 0 is sharded, every rank holds whole rows: no exchange.  row_base: this rank's first row in the
                   logical tensor and the logical row count, for the stochastic call offsets (row r draws as call
                   offset + r) -- None: equal shards, (rank * rows, world * rows); or a callable
                   `row_base(tensor) -> (first_row, total_rows)`;
      "flat"     : the tensor is this rank's contiguous piece of the flattened logical [rows, cols] tensor (an
                   FSDP-style flat shard); `logical` = (rows, cols), or a callable `logical(tensor) -> (rows, cols)`.
                   Per-row-segment maxima -> ONE all-reduce(MAX) of `rows` words -> cast with the global row scales.
    """
    global _amax_group, _amax_elem_base, _amax_layout, _amax_logical, _amax_row_base
    if layout not in (None, "channels", "flat"):
        raise ValueError("set_amax_process_group: layout must be None, 'channels' or 'flat', got %r" % (layout,))
    if layout == "flat" and logical is None:
        raise ValueError("set_amax_process_group: layout 'flat' needs logical=(rows, cols) or a callable")
    if group is None and _amax_group is not None:
        from .sharded import check_offsets
        check_offsets()   # leaving the group: the one place a sync is free
    _amax_group = group
    _amax_elem_base = elem_base
    _amax_layout = layout if group is not None else None
    _amax_logical = logical if group is not None else None
    _amax_row_base = row_base if group is not None else None


def _device_tensor(tensor):
    """True for tensors the HIP kernels can take (the CPU rehearsals of the group paths stand in for this)."""
    return tensor.is_cuda


def _shard_elem_base(tensor):
    if callable(_amax_elem_base):
        return int(_amax_elem_base(tensor))
    if _amax_elem_base is not None:
        return int(_amax_elem_base)
    return torch.distributed.get_rank(_amax_group) * tensor.numel()


def _group_rows(src, dst, fmt, rmode, mode, flt_max, rows, clamp, fine_cols):
    """Per-channel / fine-grained "max" quantisation of a tensor that is one rank's part of a logical tensor.
    rows: dim 0 of the local tensor (per-channel) or local out-channels x groups (fine-grained, fine_cols = the
    group size)."""
    if _amax_layout is None:
        raise RuntimeError("quantize_tensor: per-channel / fine-grained scaling under an amax process group needs "
                           "set_amax_process_group(group, layout='channels' | 'flat'): a row that straddles ranks "
                           "must take its maximum over all of them")
    if _amax_layout == "channels":
        offset = None
        if rmode == _hip.STOCHASTIC:
            if callable(_amax_row_base):
                first, total = _amax_row_base(src)
            else:
                world = torch.distributed.get_world_size(_amax_group)
                first, total = torch.distributed.get_rank(_amax_group) * rows, world * rows
            offset = _hip.take_offsets(int(total)) + int(first)
        _hip_quantize_rows(src, dst, fmt, rmode, flt_max, rows, clamp, offset=offset)
        return
    from .sharded import ShardedCaster
    lrows, lcols = _amax_logical(src) if callable(_amax_logical) else _amax_logical
    lrows, lcols = int(lrows), int(lcols)
    if fine_cols:   # fine-grained: every (out-channel, group) pair of the logical tensor is a row of fine_cols
        if lcols % fine_cols:
            raise NotImplementedError("fine-grained scaling of a flat shard needs equal-sized groups")
        lrows, lcols = lrows * (lcols // fine_cols), fine_cols
    ShardedCaster(_amax_group, backend=_hip).quantize_rows_flat(src, lrows, lcols, _shard_elem_base(src), mode, flt_max,
                                                                clamp=clamp, out=dst)


class TensorQuantConfig(object):
    """Quantisation configuration of one tensor role (qutils.py:22-132)."""

    def __init__(self, dtype=None, scheme=None, scaling="None", group_size=1):
        super(TensorQuantConfig, self).__init__()
        tokens = scaling.split("-")
        self.dtype = dtype
        self.scheme = scheme
        self.scaling = "tensor" in tokens
        self.per_channel = "channel" in tokens
        self.scaling_method = "mean" if "mean" in tokens else "max"
        if "int" in self.dtype:
            self.scaling_method = "none"
        self.fine_grained = scaling == "fine-grained"
        self.group_size = group_size if self.fine_grained else 1
        self.is_enabled = True

        self.dtypes_all = ["e5m2", "e4m3", "e3m4", "fp4", "bfloat16", "float16"]
        self.e5m2_modes = ["rtz", "stochastic", "rne", "rnaz", "rntz", "rpinf", "rninf",
                           "daz_stochastic", "daz_rne", "daz_rnaz", "daz_rntz"]
        self.e4m3_modes = ["rne", "stochastic", "rtz"]
        self.e4m3_ieee_modes = ["ieee_rne", "ieee_stochastic"]
        self.e4m3_all_modes = self.e4m3_modes + self.e4m3_ieee_modes
        self.e3m4_modes = ["rne", "stochastic", "rtz"]
        self.fp4_modes = ["nearest"]
        self.bfloat16_modes = ["rne", "stochastic", "rtz"]
        # the reference misspells this attribute (qutils.py:51 vs :98) so dtype "float16" raises
        # AttributeError there; here it validates.
        self.float16_modes = ["rne", "daz_rne", "stochastic"]

        self.check_validity(dtype, scheme)

    def __setattr__(self, name, value):
        # any change of the configuration drops the cached dispatch plan of quantize_tensor
        object.__setattr__(self, name, value)
        if name != "_plan":
            object.__setattr__(self, "_plan", None)

    def check_validity(self, dtype, scheme):
        if "int" in dtype:
            return  # integers of all sizes are acceptable
        assert dtype in self.dtypes_all, "Invalid data type, list of supported types : {}".format(self.dtypes_all)
        allowed = {"e5m2": self.e5m2_modes, "e4m3": self.e4m3_all_modes, "e3m4": self.e3m4_modes,
                   "fp4": self.fp4_modes, "bfloat16": self.bfloat16_modes, "float16": self.float16_modes}[dtype]
        assert scheme in allowed, "scheme {} is not in {}".format(scheme, allowed)

    def _format_key(self):
        if self.dtype == "e4m3" and self.scheme in self.e4m3_ieee_modes:
            return "e4m3_ieee"
        return self.dtype

    def get_flt_max(self):
        return _FLT_MAX.get(self._format_key())

    def get_flt_min(self):
        return _FLT_MIN.get(self.dtype)

    def __repr__(self):
        how = ("per-channel" if self.per_channel else "per-tensor" if self.scaling
               else "fine-grained" if self.fine_grained else "None")
        return "[{}, scale: {}, method: {}]".format(self.dtype + "_" + self.scheme, how, self.scaling_method)


class ModuleQuantConfig(object):
    """Quantisation configuration of one module (qutils.py:137-198)."""

    def __init__(self, wt_qconfig=None, iact_qconfig=None, oact_qconfig=None,
                 wtgrad_qconfig=None, igrad_qconfig=None, ograd_qconfig=None, patch_ops=False):
        super(ModuleQuantConfig, self).__init__()
        self.wt_qconfig = wt_qconfig
        self.iact_qconfig = iact_qconfig
        self.oact_qconfig = oact_qconfig
        self.wtgrad_qconfig = wtgrad_qconfig
        self.igrad_qconfig = igrad_qconfig
        self.ograd_qconfig = ograd_qconfig

        self.device = 'cuda'
        self.patch_ops = patch_ops
        self.patch_impl = "None"
        self.sparse_config = None
        self.global_step = -1
        self.bin_dump = False
        self.tensor_stats = False
        self.tb_writer = None
        self.weight_master = None  # master copy of the weight (bfloat16 emulation)

    def register_output_grad_hook(self, output):
        def quantize_output_grad(grad):
            if self.sparse_config is not None and self.sparse_config.outgrad:
                grad.data = self.sparse_config.sparsify_outgrad_tensor(grad.data)
            if self.ograd_qconfig is not None and self.ograd_qconfig.is_enabled:
                grad = quantize_tensor(grad, self.ograd_qconfig, None, inplace=False)
            return grad

        if output.requires_grad:
            output.register_hook(quantize_output_grad)

    def register_weight_grad_hook(self, module):
        def quantize_weight_grad(grad):
            if self.sparse_config is not None and self.sparse_config.wtgrad:
                grad.data = self.sparse_config.sparsify_wtgrad_tensor(grad.data)
            if self.wtgrad_qconfig is not None and self.wtgrad_qconfig.is_enabled:
                grad.data = quantize_tensor(grad.data, self.wtgrad_qconfig, None, inplace=True)
            return grad

        if type(module) in [torch.nn.Conv2d, torch.nn.Linear]:
            if module.weight is not None and module.weight.requires_grad:
                module.weight.register_hook(quantize_weight_grad)

    def __repr__(self):
        if self.igrad_qconfig is None:
            return "[weights: {}, inputs: {}, output: {}]".format(
                self.wt_qconfig, self.iact_qconfig, self.oact_qconfig)
        return ("[weights: {}, inputs: {}, output: {}, weight_grad: {}, input_grad: {}, "
                "output_grad: {}]").format(self.wt_qconfig, self.iact_qconfig, self.oact_qconfig,
                                           self.wtgrad_qconfig, self.igrad_qconfig, self.ograd_qconfig)


class ModuleQuantParams(object):
    """Per-module quantisation parameters; the FP formats are stateless (qutils.py:204-214)."""

    def __init__(self, wt_qparams=None, iact_qparams=None, oact_qparams=None,
                 wtgrad_qparams=None, igrad_qparams=None, ograd_qparams=None):
        super(ModuleQuantParams, self).__init__()
        self.wt_qparams = wt_qparams
        self.iact_qparams = iact_qparams
        self.oact_qparams = oact_qparams
        self.wtgrad_qparams = wtgrad_qparams
        self.igrad_qparams = igrad_qparams
        self.ograd_qparams = ograd_qparams


def quantize_to_integer(tensor, mode, inplace=False):
    """INT-n asymmetric fake quantisation in plain torch ops (qutils.py:216-247); not on the
    fpemu cast path."""
    lo, hi = torch.min(tensor), torch.max(tensor)
    nbits = int(mode.split("INT")[1]) - 1
    q_min, q_max = -1 * 2 ** nbits, (2 ** nbits) - 1
    scale = (hi - lo) / (q_max - q_min)
    zero_point = q_min - (lo / scale)
    q = torch.clamp(torch.round((tensor / scale) + zero_point), q_min, q_max)
    dq = scale * (q - zero_point)
    if inplace is True:
        tensor.data.copy_(dq)
        return tensor
    return dq


def fpemu_device_fn(tensor, mode, inplace=True, scale=1.0):
    """Device dispatch of one cast (qutils.py:249-261)."""
    if "INT" in mode:
        return quantize_to_integer(tensor, mode.split("_")[0], inplace=inplace)
    if tensor.is_cuda:
        from .pytquant import fpemu_cuda
        return fpemu_cuda.FPEmuOp.apply(tensor, mode, inplace, scale)
    from .pytquant import fpemu_cpp
    return fpemu_cpp.FPEmuOp.apply(tensor, mode, inplace, scale)


class _ScaledCast(torch.autograd.Function):
    """amax -> scale -> cast entirely on the device; straight-through backward (the same
    estimator FPEmuOp uses, pytquant/cpp/fpemu.py:68-71).  `rows` > 0 selects per-channel
    scaling over dim 0 (one scale per row of the flattened [rows, cols] view)."""

    @staticmethod
    def forward(ctx, tensor, mode, inplace, flt_max, rows, clamp, fine_cols=0):
        fmt, rmode = _hip.parse_mode(mode)
        src = tensor.contiguous()
        dst = src if inplace else torch.empty_like(src)
        if rows and _amax_group is not None and fmt not in _NO_SCALE_FMTS:
            _group_rows(src, dst, fmt, rmode, mode, flt_max, rows, clamp, fine_cols)
        elif rows:
            _hip_quantize_rows(src, dst, fmt, rmode, flt_max, rows, clamp)
        elif _amax_group is not None:
            from .sharded import all_reduce_max_words
            ws = _hip.stream_workspace(src)[:1]
            ws.zero_()
            _hip.amax_(src, ws)
            offset = _hip.next_offset(rmode)  # must advance in lock-step on every rank: verified below
            all_reduce_max_words(ws, _amax_group, check_equal=offset if rmode == _hip.STOCHASTIC else None)
            _hip.cast_scaled_(src, dst, fmt, rmode, ws, flt_max, clamp=clamp, offset=offset,
                              elem_base=_shard_elem_base(src))
        else:
            _hip.quantize_per_tensor_(src, dst, fmt, rmode, flt_max, clamp=clamp)
        if inplace:
            if src is not tensor:
                tensor.copy_(src)
            ctx.mark_dirty(tensor)
            return tensor
        return dst

    @staticmethod
    def backward(ctx, grad):
        return grad, None, None, None, None, None, None


class _KnownAmaxCast(torch.autograd.Function):
    """Per-tensor "max" quantisation of a tensor whose amax word a c-model epilogue already produced
    (_binding.note_amax): the cast alone, 8 instead of 12 bytes per element.  Same bits as _ScaledCast /
    fpemu_quantize_per_tensor (the word holds what their amax pass would find).  Straight-through backward."""

    @staticmethod
    def forward(ctx, tensor, inplace, fmt, rmode, flt_max, word):
        dst = tensor if inplace else torch.empty_like(tensor)
        _hip.cast_scaled_(tensor, dst, fmt, rmode, word, flt_max, clamp=True)
        if inplace:
            ctx.mark_dirty(tensor)
        return dst

    @staticmethod
    def backward(ctx, grad):
        return grad, None, None, None, None, None


def _hip_quantize_rows(src, dst, fmt, rmode, flt_max, rows, clamp, offset=None):
    """Per-row amax/scale/cast of a contiguous [rows, cols] buffer, no host sync.

    One launch of the rows kernel (one workgroup per row).  A few very long rows would leave most
    of the chip idle that way, so those go row by row through the full-width amax / cast kernels."""
    if fmt in (_hip.FMT_BF16, _hip.FMT_FP16, _hip.FMT_FP16_DAZ):
        _hip.cast_(src, dst, fmt, rmode)  # these formats take no scale (cpp/fpemu_impl.cpp:1933-1957)
        return
    cols = src.numel() // rows
    if rows >= 64 or cols <= (1 << 16):
        _hip.quantize_rows_(src, dst, rows, fmt, rmode, flt_max, clamp=clamp, offset=offset)
        return
    s2, d2 = src.view(rows, -1), dst.view(rows, -1)
    ws = torch.zeros(rows, dtype=torch.int32, device=src.device)
    for r in range(rows):
        _hip.amax_(s2[r], ws[r:r + 1])
    for r in range(rows):
        _hip.cast_scaled_(s2[r], d2[r], fmt, rmode, ws[r:r + 1], flt_max, clamp=clamp,
                          offset=None if offset is None else offset + r)


def _device_mean_scales(stat, flt_min, threshold):
    """float32 scales from the statistics `stat` (one element per scale) by the reference's rule
    (qutils.py:278-281 / :298-301), as elementwise torch ops on the device -- the same operations, in the same
    order and dtype, that the reference applies to its 0-d tensor (`flt_min / tensor` is torch's
    reciprocal-then-multiply in both), so each element equals the scale the host-side loop would compute."""
    a = abs(stat)
    big = a > threshold
    scale = flt_min / a
    scale = torch.where(big, scale, 1.0)       # below the threshold (or NaN): mean := flt_min, scale = 1
    scale = torch.where(scale < 1.0, 1.0, scale)
    return scale.float().contiguous()


class _GivenScaleCast(torch.autograd.Function):
    """Cast with scale(s) already in device memory: `rows` = 0 one scale for the tensor, else one per row of the
    [rows, cols] view.  Straight-through backward, as FPEmuOp (pytquant/cpp/fpemu.py:68-71)."""

    @staticmethod
    def forward(ctx, tensor, mode, inplace, scales, rows):
        fmt, rmode = _hip.parse_mode(mode)
        src = tensor.contiguous()
        dst = src if inplace else torch.empty_like(src)
        if fmt in _NO_SCALE_FMTS:
            # these formats take no scale (cpp/fpemu_impl.cpp:1933-1957).  (Not reachable through quantize_tensor:
            # they have no flt_min, so the "mean" scale rule raises first -- as in the reference, qutils.py:116-129.)
            _hip.cast_(src, dst, fmt, rmode)
        elif not rows:
            _hip.cast_device_scale_(src, dst, fmt, rmode, scales)
        elif rows >= 64 or src.numel() // rows <= (1 << 16):
            _hip.cast_rows_scaled_(src, dst, rows, fmt, rmode, scales)
        else:  # a few very long rows: full-width launches, row by row (see _hip_quantize_rows)
            s2, d2 = src.view(rows, -1), dst.view(rows, -1)
            for r in range(rows):
                _hip.cast_device_scale_(s2[r], d2[r], fmt, rmode, scales[r:r + 1])
        if inplace:
            if src is not tensor:
                tensor.copy_(src)
            ctx.mark_dirty(tensor)
            return tensor
        return dst

    @staticmethod
    def backward(ctx, grad):
        return grad, None, None, None, None


def _mean_scale(sample, qtconfig, threshold, use_mode):
    """The reference's "mean" policies, host-synchronising like the original
    (qutils.py:276-281 per tensor; :294-301 / :322-327 per channel / group use torch.mode).  Used where the
    device-side forms above do not apply (CPU tensors for the INT modes, ragged fine-grained groups)."""
    flat = torch.flatten(sample.detach())
    if use_mode:
        mean = abs(torch.mode(flat).values.data)
    else:
        mean = torch.mean(abs(flat))
    flt_min = qtconfig.get_flt_min()
    mean = abs(mean) if abs(mean) > threshold else flt_min
    scale = 1.0
    if abs(mean) > 0.0:
        scale = flt_min / abs(mean)
    return 1.0 if scale < 1.0 else scale


def _plan(qtconfig):
    """(kind, mode, fmt, rmode, flt_max) of a configuration, computed once per configuration state (the hooks
    call quantize_tensor tens to hundreds of times per step with the same few configurations):
    kind 1 = per-tensor max scaling, 2 = plain cast, 3 = per-channel max scaling, 0 = everything else (the general
    code below)."""
    qtconfig.check_validity(qtconfig.dtype, qtconfig.scheme)
    mode = qtconfig.dtype.upper()
    if qtconfig.scheme is not None:
        mode += "_" + qtconfig.scheme.upper()
    kind, fmt, rmode = 0, -1, -1
    if "INT" not in mode:
        fmt, rmode = _hip.parse_mode(mode)
        if qtconfig.scaling is True:
            kind = 1 if (qtconfig.scaling_method == "max" and fmt not in _NO_SCALE_FMTS) else 0
        elif qtconfig.per_channel is True:
            kind = 3 if (qtconfig.scaling_method == "max" and fmt not in _NO_SCALE_FMTS) else 0
        elif not qtconfig.fine_grained:
            kind = 2
    plan = (kind, mode, fmt, rmode, qtconfig.get_flt_max())
    qtconfig._plan = plan
    return plan


class _Memo(threading.local):
    """The last out-of-place deterministic quantisation on this thread: (input, its data_ptr / _version, plan, output,
    the output's _version).  Modules that share an input -- the q / k / v projections of an attention block, a
    bottleneck's first convolution and its downsample branch -- have the SAME tensor quantised with the SAME
    configuration by consecutive forward-pre hooks (reference qutils.py:528-534, one quantize_tensor per module): a pure
    function of an unchanged input, so the second and third calls return the first one's result instead of two more
    amax passes and casts.  The entry holds both tensors (an address cannot be reused while it lives) and is dropped on
    any mismatch: another tensor, an in-place change of the input (its _version moved) or of the result (a consumer
    wrote into it), another configuration, another autograd state.  Stochastic modes never enter (every call draws its
    own noise), nor do calls under stream capture.  Only calls made inside `shared_input_memo()` take part -- the
    forward-pre hook's quantisation of a module's INPUTS: activations are real autograd tensors, whose in-place changes
    move their _version, whereas weights are reached through `.data` aliases (a fresh version counter per access: a
    `weight.data.copy_(...)` between two calls would go unseen).  Cost: the last input / result pair stays alive until
    the next such quantisation on the thread.  Measured: BERT-base 8 x 128 with E4M3 PTQ hooks 3.93 -> 3.85 ms (24 of its
    148 casts per forward are repeats).  MPEMU_NO_MEMO=1 switches it off."""
    slot = None
    depth = 0     # > 0: inside shared_input_memo()


_memo = _Memo()
_MEMO_ON = os.environ.get("MPEMU_NO_MEMO", "0") in ("", "0")


class shared_input_memo(object):
    """Context: out-of-place deterministic quantize_tensor calls on this thread may reuse / feed the one-entry memo."""

    def __enter__(self):
        _memo.depth += 1
        return self

    def __exit__(self, *a):
        _memo.depth -= 1
        return False


def _memo_lookup(tensor, plan):
    m = _memo.slot
    if (m is not None and m[1] == tensor.data_ptr() and m[2] == tensor._version and m[3] == plan
            and m[0].shape == tensor.shape and m[0].dtype == tensor.dtype and m[0].device == tensor.device
            and m[0].stride() == tensor.stride() and m[5] == m[4]._version
            and m[6] == (tensor.requires_grad and torch.is_grad_enabled())):
        return m[4]
    return None


def _memo_invalidate(tensor):
    """An in-place quantisation of `tensor` is about to run: drop the entry if it lives in the same storage as the
    entry's input or result (the native in-place paths write through raw pointers and `.data` aliases, which move no
    _version the entry could see)."""
    m = _memo.slot
    if m is not None:
        sp = tensor.untyped_storage().data_ptr()
        if sp == m[0].untyped_storage().data_ptr() or sp == m[4].untyped_storage().data_ptr():
            _memo.slot = None


def _memo_store(tensor, plan, out):
    _memo.slot = (tensor, tensor.data_ptr(), tensor._version, plan, out, out._version,
                  tensor.requires_grad and torch.is_grad_enabled())


def quantize_tensor(tensor, qtconfig, qtparams, inplace=False):
    """Single entry point for quantising a tensor (qutils.py:264-345)."""
    plan = getattr(qtconfig, "_plan", None) or _plan(qtconfig)
    if inplace and _memo.slot is not None:
        _memo_invalidate(tensor)
    f64 = tensor.dtype == torch.float64   # plain casts only on the device; scales are computed as the reference does
    if plan[0] == 3:
        # per-channel "max": one native call (checks, allocation, the rows kernel); the reference's loop over dim 0
        # never goes through autograd for the pieces (qutils.py:289-311), so there is no node to build
        ext = _hip.ext()
        rows = tensor.shape[0] if tensor.dim() >= 1 else 0
        if (ext is not None and rows > 0 and _amax_group is None and tensor.is_cuda and not tensor.is_sparse and not f64
                and not _hip.tracing() and (rows >= 64 or tensor.numel() // rows <= (1 << 16))):
            q = ext.quantize_rows(tensor.detach(), bool(inplace), plan[2], plan[3], plan[4], rows, True,
                                  _hip.get_rng_state()[0],
                                  _hip.take_offsets(rows) if plan[3] == _hip.STOCHASTIC else 0)
            return tensor if inplace else q
    elif plan[0] and _amax_group is None and tensor.is_cuda and not tensor.is_sparse and not (f64 and plan[0] == 1):
        if plan[0] == 1 and _hip.has_known_amax():
            word = _hip.take_amax(tensor)   # left by the c-model op that produced this very tensor
            if word is not None:
                return _KnownAmaxCast.apply(tensor, bool(inplace), plan[2], plan[3], plan[4], word)
        ext = _hip.ext()
        if ext is not None and not _hip.tracing():
            # the two cases every hook call is: one native call does checks, allocation, launch(es) and the
            # straight-through autograd node (csrc/mpemu_torch.cpp)
            # (never under stream capture: a captured call's result holds nothing until the graph is replayed)
            memo = (_memo.depth > 0 and _MEMO_ON and not inplace and plan[3] != _hip.STOCHASTIC
                    and not torch.cuda.is_current_stream_capturing())
            if memo:
                hit = _memo_lookup(tensor, plan)
                if hit is not None:
                    return hit
            if plan[0] == 1:
                out = ext.quantize_per_tensor(tensor, bool(inplace), plan[2], plan[3], plan[4], True,
                                              _hip.get_rng_state()[0], _hip.next_offset(plan[3]),
                                              _hip.workspace_override())
            else:
                out = ext.fpemu_op(tensor, bool(inplace), plan[2], plan[3], 1.0, _hip.get_rng_state()[0],
                                   _hip.next_offset(plan[3]))
            if memo:
                _memo_store(tensor, plan, out)
            return out
    qtconfig.check_validity(qtconfig.dtype, qtconfig.scheme)
    mode = qtconfig.dtype.upper()
    if qtconfig.scheme is not None:
        mode += "_" + qtconfig.scheme.upper()
    # float64 tensors (the reference's GPU dispatcher takes them, hip/fpemu_kernels.hip:1070): the statistic and the
    # scale are computed by torch in double and handed over as a float, exactly as the reference's Python does
    # (qutils.py:276-287); the device-side scale paths are float32 / 16-bit only
    on_device = _device_tensor(tensor) and "INT" not in mode and not f64

    if qtconfig.scaling is True:
        if qtconfig.scaling_method == "max" and on_device:
            ext = _hip.ext()
            if ext is not None and _amax_group is None and not _hip.tracing():
                # amax -> device-side scale -> cast and the straight-through autograd node in one native call
                fmt, rmode = _hip.parse_mode(mode)
                if fmt not in _NO_SCALE_FMTS:
                    return ext.quantize_per_tensor(tensor, bool(inplace), fmt, rmode, qtconfig.get_flt_max(), True,
                                                   _hip.get_rng_state()[0], _hip.next_offset(rmode),
                                                   _hip.workspace_override())
            return _ScaledCast.apply(tensor, mode, inplace, qtconfig.get_flt_max(), 0, True)
        scale = 1.0
        if qtconfig.scaling_method == "mean":
            if on_device and tensor.numel() > 0:
                if _amax_group is not None:
                    raise NotImplementedError("per-tensor-mean scaling of a sharded tensor: the mean is not reduced "
                                              "over the amax process group")
                # the statistic and the scale rule stay torch ops on the device; the cast reads the scale there
                mean = torch.mean(abs(torch.flatten(tensor.detach())))
                scales = _device_mean_scales(mean.reshape(1), qtconfig.get_flt_min(), 1e-5)
                return _GivenScaleCast.apply(tensor, mode, inplace, scales, 0)
            scale = _mean_scale(tensor, qtconfig, 1e-5, use_mode=False)
        elif qtconfig.scaling_method == "max":
            vmax = torch.max(abs(torch.flatten(tensor.detach())))
            scale = qtconfig.get_flt_max() / vmax
            scale = 6.55e+04 if scale > 3.275e+04 else scale
        return fpemu_device_fn(tensor, mode, inplace=inplace, scale=scale)

    if qtconfig.per_channel is True:
        if qtconfig.scaling_method == "max" and on_device and tensor.dim() >= 1 and tensor.shape[0] > 0:
            # the reference loops over dim 0 and never goes through autograd for the pieces
            q = _ScaledCast.apply(tensor.detach(), mode, inplace, qtconfig.get_flt_max(), tensor.shape[0], True)
            return tensor if inplace else q
        if (qtconfig.scaling_method == "mean" and on_device and tensor.dim() >= 1 and tensor.shape[0] > 0
                and tensor.numel() > 0):
            # torch.mode of every row in one call, the scale rule elementwise, one cast launch for all rows
            if _amax_group is not None and _amax_layout != "channels":
                raise NotImplementedError("per-channel-mean scaling of a sharded tensor: the row statistic is not "
                                          "reduced over the amax process group (layout 'channels' needs no exchange)")
            rows = tensor.shape[0]
            flat = tensor.detach().contiguous().view(rows, -1)
            scales = _device_mean_scales(torch.mode(flat, dim=1).values, qtconfig.get_flt_min(), 1e-6)
            q = _GivenScaleCast.apply(tensor.detach(), mode, inplace, scales, rows)
            return tensor if inplace else q
        tensor_q = None if inplace else torch.zeros_like(tensor)
        for c in range(tensor.shape[0]):
            sub = tensor.select(0, c).detach()
            scale = 1.0
            if qtconfig.scaling_method == "mean":
                scale = _mean_scale(sub, qtconfig, 1e-6, use_mode=True)
            elif qtconfig.scaling_method == "max":
                vmax = torch.max(abs(torch.flatten(sub)))
                scale = qtconfig.get_flt_max() / vmax
                scale = 6.55e+04 if scale > 3.275e+04 else scale
            sub = fpemu_device_fn(sub, mode, inplace=False, scale=scale)
            (tensor if inplace else tensor_q).select(0, c).data.copy_(sub)
        return tensor if inplace else tensor_q

    if qtconfig.fine_grained is True:
        if _amax_group is not None and _amax_layout == "flat" and qtconfig.scaling_method == "max" and on_device:
            # a flat shard carries no shape: the groups come from the logical [out, in] tensor
            lcols = int((_amax_logical(tensor) if callable(_amax_logical) else _amax_logical)[1])
            chunks = max(1, int(lcols / qtconfig.group_size))
            size = -(-lcols // chunks)
            q = _ScaledCast.apply(tensor.detach(), mode, inplace, qtconfig.get_flt_max(), 1, False, size)
            return tensor if inplace else q
        if (qtconfig.scaling_method == "max" and on_device and tensor.dim() >= 2 and tensor.is_contiguous()
                and tensor.numel() > 0):
            # groups are torch.chunk pieces of dim 1; when they are equal-sized every (out-channel,
            # group) pair is one contiguous row and the whole tensor is one rows-kernel launch
            in_ch = tensor.shape[1]
            chunks = max(1, int(in_ch / qtconfig.group_size))
            size = -(-in_ch // chunks)
            if in_ch % size == 0:
                rows = tensor.shape[0] * (in_ch // size)
                q = _ScaledCast.apply(tensor.detach(), mode, inplace, qtconfig.get_flt_max(), rows, False, size)
                return tensor if inplace else q
            if _amax_group is not None:
                raise NotImplementedError("fine-grained scaling with ragged groups under an amax process group")
        tensor_q = None if inplace else torch.zeros_like(tensor)
        for k in range(tensor.shape[0]):
            crs = tensor.select(0, k).detach()
            if not inplace:
                crs = crs.clone()
            chunks = max(1, int(crs.shape[0] / qtconfig.group_size))
            for sub in crs.chunk(chunks, 0):
                if qtconfig.scaling_method == "max" and on_device and sub.is_contiguous():
                    _ScaledCast.apply(sub, mode, True, qtconfig.get_flt_max(), 0, False)
                    continue
                scale = 1.0
                if qtconfig.scaling_method == "mean":
                    scale = _mean_scale(sub, qtconfig, 1e-6, use_mode=True)
                elif qtconfig.scaling_method == "max":
                    vmax = torch.max(abs(torch.flatten(sub)))
                    scale = qtconfig.get_flt_max() / vmax
                fpemu_device_fn(sub, mode, inplace=True, scale=scale)
            if not inplace:
                tensor_q.select(0, k).data.copy_(crs)
        return tensor if inplace else tensor_q

    return fpemu_device_fn(tensor, mode, inplace=inplace)


# ---------------------------------------------------------------------------------------------
# Model-level configuration and the module hooks (reference qutils.py:351-365, 458-475, 478-660).
# The hooks only decide WHICH tensors are cast with WHICH config; every cast goes through
# quantize_tensor above, i.e. the HIP kernels.  Out of scope here (not on the cast path): the
# stats collectors, tensor dumps and tensorboard plumbing of the reference (flags are accepted
# and ignored).  Op patching (`patch_ops`, patch_impl "SIMPLE"): for the duration of a hooked module's
# forward, torch.addmm / matmul / mm (functions and Tensor methods) and torch.nn.functional.conv2d are
# replaced by the c-model ops of mpemu.cmodel.simple, as the reference does for CPU modules
# (qutils.py:480-509, 588-600); here the c-model runs on the GPU, so it applies to GPU modules.
# ---------------------------------------------------------------------------------------------
def get_or_update_model_quant_config_dict(model, filter_module_types, mod_qconfig, prefix="",
                                          model_qconfig_dict=None, override=False, exempt_modules=[]):
    """name -> deep copy of mod_qconfig for every module whose exact type is listed."""
    if model_qconfig_dict is None:
        model_qconfig_dict = OrderedDict()
    for name, module in model.named_modules():
        fname = prefix + name
        listed = type(module) in filter_module_types
        if listed and fname not in exempt_modules:
            if override or fname not in model_qconfig_dict:
                model_qconfig_dict[fname] = copy.deepcopy(mod_qconfig)
        elif not listed and fname in model_qconfig_dict:
            model_qconfig_dict.pop(fname)
    return model_qconfig_dict


def get_quantization_parameters(tensor, qtconfig):
    """The floating-point formats are stateless: no parameters (qutils.py:448-456)."""
    return None


def quantize_weights_of_a_module(module, wt_qconfig, name=""):
    module.weight.data.copy_(quantize_tensor(module.weight.data, wt_qconfig, None))


def quantize_model_weights(model, model_qconfig_dict):
    """The reference's loop of quantize_weights_of_a_module over the model (qutils.py:470-475).  Weights whose
    configuration is per-tensor "max" scaling with a deterministic rounding mode -- what every emulator preset
    gives Linear / Conv2d weights outside the per-channel presets -- are collected per (configuration, dtype,
    device) and quantised in place by fpemu_quantize_per_tensor_multi: two launches per 32 weights instead of one to
    three per weight, bit for bit the loop's result (tensor i of a call is quantised exactly as the single-tensor
    call would; deterministic modes consume no call offsets, so the order of the calls does not matter).
    Under an amax process group (every weight a shard of a logical weight) the collection goes amax words ->
    ONE all-reduce(MAX) of all the words -> multi-tensor cast, instead of one collective per weight.
    A storage that several modules share (tied weights) enters the batch once; its later occurrences are quantised
    after the batch, one by one, so the tensor ends as q(q(w)) exactly as in the reference's sequential loop.
    Everything else takes the loop."""
    batches = OrderedDict()
    seen, again = set(), []
    for name, module in model.named_modules():
        cfg = model_qconfig_dict.get(name)
        if cfg is None or cfg.wt_qconfig is None:
            continue
        qt = cfg.wt_qconfig
        w = module.weight.data
        plan = getattr(qt, "_plan", None) or _plan(qt)
        if (plan[0] == 1 and plan[3] != _hip.STOCHASTIC and _device_tensor(w) and w.is_contiguous() and not w.is_sparse
                and w.dtype in (torch.float32, torch.float16, torch.bfloat16)):
            key = (w.data_ptr(), w.numel())
            if key in seen:
                again.append((module, qt, name))
                continue
            seen.add(key)
            batches.setdefault((plan[2], plan[3], plan[4], w.dtype, w.device), []).append(w)
        else:
            quantize_weights_of_a_module(module, qt, name=name)
    for (fmt, rmode, flt_max, _, _), weights in batches.items():
        if _amax_group is not None:
            from .sharded import ShardedCaster
            ShardedCaster(_amax_group, backend=_hip).quantize_per_tensor_multi(weights, (fmt, rmode), flt_max)
        elif len(weights) == 1:
            _hip.quantize_per_tensor_(weights[0], weights[0], fmt, rmode, flt_max)
        else:
            _hip.quantize_per_tensor_multi_(weights, fmt, rmode, flt_max)
    for module, qt, name in again:
        quantize_weights_of_a_module(module, qt, name=name)


def _active(qtconfig):
    return qtconfig is not None and qtconfig.is_enabled


def quantize_module_weights_and_inputs(self, input):
    """forward-pre hook: weights in place, then every positional input out of place."""
    qc = self.qconfig
    if qc.patch_ops and qc.patch_impl == "SIMPLE" and any(True for _ in self.parameters()):
        _patch_ops(self)
    keeps_master = qc.wt_qconfig is not None and (qc.wt_qconfig.dtype in ["bfloat16"] or qc.device == 'cpu')
    if keeps_master:
        qc.weight_master = copy.deepcopy(self.weight.data)
    if qc.sparse_config is not None and qc.sparse_config.weight and type(self) in [torch.nn.Conv2d, torch.nn.Linear]:
        self.weight.data = qc.sparse_config.sparsify_weight_tensor(self.weight.data)
    if _active(qc.wt_qconfig):
        self.weight.data = quantize_tensor(self.weight.data, qc.wt_qconfig, self.qparams.wt_qparams, inplace=True)
    if _active(qc.iact_qconfig):
        with shared_input_memo():   # modules that share an input: the repeat of the last quantisation is its result
            input = tuple(quantize_tensor(t, qc.iact_qconfig, None, inplace=False) for t in input)
    return input


def quantize_module_gradient(self, grad_input, grad_output):
    """full-backward hook: input gradients out of place; restores a master weight copy if kept."""
    qc = self.qconfig
    if qc.sparse_config is not None and qc.sparse_config.ingrad:
        grad_input = tuple(qc.sparse_config.sparsify_ingrad_tensor(g) for g in grad_input)
    if _active(qc.igrad_qconfig):
        grad_input = tuple(g if g is None else quantize_tensor(g, qc.igrad_qconfig, self.qparams.igrad_qparams,
                                                                inplace=False) for g in grad_input)
    if qc.wt_qconfig is not None and (qc.wt_qconfig.dtype in ["bfloat16"] or qc.device == 'cpu'):
        self.weight.data = copy.deepcopy(qc.weight_master)
    return grad_input


_PATCHED = (("addmm", "addmm"), ("matmul", "matmul"), ("mm", "mm"))


def _patch_ops(module):
    """Swap in the c-model ops (reference qutils.py:489-509); undone by the module's forward hook."""
    from .cmodel import simple as hw_impl
    if getattr(module, "_mpemu_saved_ops", None) is not None:
        return
    saved = {}
    for tname, sname in _PATCHED:
        saved["torch." + tname] = getattr(torch, tname)
        saved["Tensor." + tname] = getattr(torch.Tensor, tname)
        setattr(torch, tname, getattr(hw_impl, sname))
        setattr(torch.Tensor, tname, getattr(hw_impl, sname))
    saved["F.conv2d"] = torch.nn.functional.conv2d
    torch.nn.functional.conv2d = hw_impl.conv2d
    # the output is quantised per tensor right after the forward: let the c-model's epilogue find its maximum
    oq = module.qconfig.oact_qconfig
    if _active(oq) and _amax_group is None and (getattr(oq, "_plan", None) or _plan(oq))[0] == 1:
        saved["collect"] = hw_impl.collect_output_amax(True)
    module._mpemu_saved_ops = saved


def _unpatch_ops(module):
    saved = getattr(module, "_mpemu_saved_ops", None)
    if saved is None:
        return
    for tname, _ in _PATCHED:
        setattr(torch, tname, saved["torch." + tname])
        setattr(torch.Tensor, tname, saved["Tensor." + tname])
    torch.nn.functional.conv2d = saved["F.conv2d"]
    if "collect" in saved:
        from .cmodel import simple as hw_impl
        hw_impl.collect_output_amax(saved["collect"])
    module._mpemu_saved_ops = None


def quantize_module_output(self, input, output):
    """forward hook: restores patched ops, registers the output-gradient hook, then casts the output in
    place."""
    qc = self.qconfig
    _unpatch_ops(self)
    qc.register_output_grad_hook(output)
    if _active(qc.oact_qconfig):
        output = quantize_tensor(output, qc.oact_qconfig, self.qparams.oact_qparams, inplace=True)
    _hip.forget_amax()   # a maximum nobody asked for does not outlive its module
    return output


def add_quantization_hooks(model, model_qconfig_dict, is_training=False):
    handles = []
    for name, module in model.named_modules():
        if name not in model_qconfig_dict:
            continue
        handles.append(module.register_forward_pre_hook(quantize_module_weights_and_inputs))
        handles.append(module.register_forward_hook(quantize_module_output))
        if is_training:
            if module.qconfig.igrad_qconfig is not None:
                handles.append(module.register_full_backward_hook(quantize_module_gradient))
            module.qconfig.register_weight_grad_hook(module)
    return handles


def reset_quantization_setup(model, model_qconfig_dict):
    """Attach .name / .qconfig / .qparams to every configured module."""
    for name, module in model.named_modules():
        if name in model_qconfig_dict:
            module.name = name
            module.qconfig = copy.deepcopy(model_qconfig_dict[name])
            module.qparams = ModuleQuantParams()


def _set_flag(model, model_qconfig_dict, attr, value):
    for name, module in model.named_modules():
        if name in model_qconfig_dict:
            cfg = getattr(module.qconfig, attr)
            if cfg is not None:
                cfg.is_enabled = value


def set_quantize_weights_flag(model, model_qconfig_dict, set_value):
    _set_flag(model, model_qconfig_dict, "wt_qconfig", set_value)


def set_quantize_outputs_flag(model, model_qconfig_dict, set_value):
    _set_flag(model, model_qconfig_dict, "oact_qconfig", set_value)


def set_quantize_inputs_flag(model, model_qconfig_dict, set_value):
    _set_flag(model, model_qconfig_dict, "iact_qconfig", set_value)
