"""ctypes binding of libfpemu_hip.so (C ABI: include/fpemu_hip.h).

This is the thin host layer between torch tensors and the hand-written gfx950 kernels.  torch
supplies device memory and the current HIP stream only; no arithmetic happens in torch here.
There is NO fallback: if the library is missing or the tensor is not on a ROCm device the call
raises.

Stands where the reference's pybind module `fpemu_hip` stands
(mpemu/pytquant/hip/fpemu_impl.cpp:27-53): `forward(...)` below keeps that signature.
"""
import ctypes
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libfpemu_hip.so")

F32, F16, BF16, F64 = 0, 1, 2, 3
# float64 storage: plain casts with a host-side scale (fpemu_cast / fpemu_forward); the library refuses it elsewhere
_DTYPES = {torch.float32: F32, torch.float16: F16, torch.bfloat16: BF16, torch.float64: F64}

# format ids / flt_max of include/fpemu_hip.h
FMT_E5M2, FMT_E4M3, FMT_E3M4, FMT_BF16, FMT_FP16, FMT_E5M2_DAZ, FMT_FP16_DAZ = range(7)
STOCHASTIC = 2

_ERR = {-1: ValueError, -2: ValueError, -3: NotImplementedError, -4: ValueError}

_lib = None
_lock = threading.Lock()


def lib():
    """Load libfpemu_hip.so; raise loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.isfile(_SO):
            raise ImportError(
                "libfpemu_hip.so not found at %s: build it with `make -C fp8-emulation-toolkit_amd/csrc` "
                "(or __graft_entry__.build()); there is no CPU or eager fallback" % _SO)
        L = ctypes.CDLL(_SO)
        u64, i32, f32, vp = ctypes.c_uint64, ctypes.c_int, ctypes.c_float, ctypes.c_void_p
        L.fpemu_abi_version.restype = i32
        L.fpemu_last_error.restype = ctypes.c_char_p
        L.fpemu_parse_mode.restype = i32
        L.fpemu_parse_mode.argtypes = [ctypes.c_char_p, ctypes.POINTER(i32), ctypes.POINTER(i32)]
        L.fpemu_cast.restype = i32
        L.fpemu_cast.argtypes = [vp, vp, u64, i32, i32, i32, f32, u64, u64, u64, vp]
        L.fpemu_forward.restype = i32
        L.fpemu_forward.argtypes = [vp, vp, i32, ctypes.c_char_p, u64, f32, i32, i32, u64, u64, vp]
        L.fpemu_amax.restype = i32
        L.fpemu_amax.argtypes = [vp, u64, i32, vp, vp]
        L.fpemu_cast_scaled.restype = i32
        L.fpemu_cast_scaled.argtypes = [vp, vp, u64, i32, i32, i32, vp, f32, i32, u64, u64, u64, vp]
        L.fpemu_quantize_per_tensor.restype = i32
        L.fpemu_quantize_per_tensor.argtypes = [vp, vp, u64, i32, i32, i32, f32, i32, vp, u64, u64, vp]
        L.fpemu_quantize_per_tensor_ws.restype = i32
        L.fpemu_quantize_per_tensor_ws.argtypes = [vp, vp, u64, i32, i32, i32, f32, i32, vp, u64, u64, u64, vp]
        L.fpemu_quantize_per_tensor_static.restype = i32
        L.fpemu_quantize_per_tensor_static.argtypes = [vp, vp, u64, i32, i32, i32, f32, i32, vp, u64, u64, vp]
        L.fpemu_cast_device_scale.restype = i32
        L.fpemu_cast_device_scale.argtypes = [vp, vp, u64, i32, i32, i32, vp, u64, u64, u64, vp]
        L.fpemu_cast_rows_scaled.restype = i32
        L.fpemu_cast_rows_scaled.argtypes = [vp, vp, u64, u64, i32, i32, i32, vp, u64, u64, vp]
        L.fpemu_quantize_rows.restype = i32
        L.fpemu_quantize_rows.argtypes = [vp, vp, u64, u64, i32, i32, i32, f32, i32, u64, u64, vp]
        i64 = ctypes.c_int64
        L.fpemu_gemm_f32.restype = i32
        L.fpemu_gemm_f32.argtypes = [vp, vp, vp, i64, i64, i64, i64, i64, i64, f32, i32, i32, i32, vp]
        L.fpemu_gemm_f32_amax.restype = i32
        L.fpemu_gemm_f32_amax.argtypes = [vp, vp, vp, i64, i64, i64, i64, i64, i64, f32, i32, i32, i32, vp, vp]
        L.fpemu_gemm_chain_f32.restype = i32
        L.fpemu_gemm_chain_f32.argtypes = [vp, vp, vp] + [i64] * 6 + [i32, i32, i32] + [i64] * 7 + [vp]
        L.fpemu_conv2d_workspace_bytes.restype = u64
        L.fpemu_conv2d_workspace_bytes.argtypes = [i32] + [i64] * 9
        L.fpemu_conv2d_fp_f32.restype = i32
        L.fpemu_conv2d_fp_f32.argtypes = [vp, vp, vp, vp] + [i64] * 9 + [i32, vp, u64, vp]
        L.fpemu_conv2d_fp_f32_amax.restype = i32
        L.fpemu_conv2d_fp_f32_amax.argtypes = [vp, vp, vp, vp] + [i64] * 9 + [i32, vp, u64, vp, vp]
        for nm in ("fpemu_conv2d_bp_f32", "fpemu_conv2d_wu_f32"):
            f = getattr(L, nm)
            f.restype = i32
            f.argtypes = [vp, vp, vp] + [i64] * 9 + [i32, vp, u64, vp]
        L.fpemu_test_set_max_blocks_.restype = None   # test hook, not part of the public header
        L.fpemu_test_set_max_blocks_.argtypes = [i32]
        L.fpemu_quantize_blocks.restype = i32
        L.fpemu_quantize_blocks.argtypes = [vp, vp, u64, u64, i32, i32, i32, u64, u64, vp]
        L.fpemu_cast_multi.restype = i32
        L.fpemu_cast_multi.argtypes = [vp, u64, i32, i32, i32, f32, u64, u64, vp]
        L.fpemu_quantize_per_tensor_multi.restype = i32
        L.fpemu_quantize_per_tensor_multi.argtypes = [vp, u64, i32, i32, i32, f32, i32, vp, u64, u64, u64, vp]
        L.fpemu_amax_multi.restype = i32
        L.fpemu_amax_multi.argtypes = [vp, u64, i32, vp, vp]
        L.fpemu_cast_multi_scaled.restype = i32
        L.fpemu_cast_multi_scaled.argtypes = [vp, u64, i32, i32, i32, vp, f32, i32, u64, u64, vp]
        L.fpemu_amax_rows.restype = i32
        L.fpemu_amax_rows.argtypes = [vp, u64, u64, i32, vp, vp]
        L.fpemu_cast_rows_amax.restype = i32
        L.fpemu_cast_rows_amax.argtypes = [vp, vp, u64, u64, i32, i32, i32, vp, f32, i32, u64, u64, u64, vp]
        if L.fpemu_abi_version() != 10:
            raise ImportError("libfpemu_hip.so: ABI version mismatch")
        _lib = L
    return _lib


_ext = False  # not looked up yet


def ext():
    """The thin PyTorch-ROCm extension (_mpemu_torch.so, csrc/mpemu_torch.cpp) over the same C ABI: the
    operator's two hot calls with checks, allocation, stream lookup and the straight-through autograd node in
    C++.  None when it has not been built (or MPEMU_NO_TORCH_EXT=1): callers then take the ctypes path below --
    the same library, the same bits, more host time per call."""
    global _ext
    if _ext is False:
        _ext = None
        if os.environ.get("MPEMU_NO_TORCH_EXT", "0") in ("", "0"):
            lib()  # libfpemu_hip.so first: the extension links against it
            try:
                from . import _mpemu_torch
                if _mpemu_torch.abi_version() == lib().fpemu_abi_version():
                    _ext = _mpemu_torch
            except ImportError:
                _ext = None
    return _ext


def workspace_override():
    """The scratch installed by use_workspace() on this thread, or None."""
    return getattr(_tls, "ws", None)


def tracing():
    return _trace


def _check(rc):
    if rc == 0:
        return
    msg = lib().fpemu_last_error().decode(errors="replace")
    if rc < 0:
        raise _ERR.get(rc, RuntimeError)(msg)
    raise RuntimeError("HIP error %d: %s" % (rc, msg))


def parse_mode(mode):
    """(fmt, rmode) ids of a reference mode string; raises ValueError for an unknown string
    (the reference silently does nothing, cpp/fpemu_impl.cpp:1994-2000)."""
    hit = _mode_cache.get(mode)
    if hit is not None:
        return hit
    f, r = ctypes.c_int(), ctypes.c_int()
    _check(lib().fpemu_parse_mode(str(mode).encode(), ctypes.byref(f), ctypes.byref(r)))
    _mode_cache[mode] = (f.value, r.value)
    return f.value, r.value


def _dtype_id(t):
    try:
        return _DTYPES[t.dtype]
    except KeyError:
        raise TypeError("fpemu: unsupported tensor dtype %s (float32, float64, float16, bfloat16 only)" % t.dtype)


def _require_device(t, what="input"):
    if not t.is_cuda:
        raise RuntimeError(
            "fpemu: %s must live on a ROCm device (got %s); this build has no CPU path" % (what, t.device))
    if not t.is_contiguous():
        raise RuntimeError("fpemu: %s must be contiguous" % what)  # hip/fpemu_impl.cpp:23-25


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream(t):
    """Raw hipStream_t of torch's current stream on t's device (the launch stream)."""
    if _raw_stream is not None:
        return _raw_stream(t.device.index)
    return torch.cuda.current_stream(t.device).cuda_stream


class _NoCtx(object):
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


_NOCTX = _NoCtx()


def _on_device(t):
    """Context that makes t's device current; free when it already is (the common case)."""
    if _cur_device is not None and _cur_device() == t.device.index:
        return _NOCTX
    return torch.cuda.device_of(t)


_mode_cache = {}

# ---- optional profiler ranges (the reference marks its c-model ops with RECORD_FUNCTION,
# cmodel/simple/simple_gemm.cpp:47; here every native entry can carry one) ---------------------
_trace = os.environ.get("MPEMU_TRACE", "0") not in ("", "0")


def set_tracing(on):
    """Wrap each native call in a torch.profiler.record_function range (also MPEMU_TRACE=1)."""
    global _trace
    _trace = bool(on)


def _range(name):
    return torch.profiler.record_function(name) if _trace else _NOCTX


# ---- stochastic-rounding stream state: (seed, per-call offset) -----------------------------
_seed = 0x5EED0F8E3A11
_offset = 0


def manual_seed(seed):
    """Re-key the Philox stream of the stochastic modes and rewind the call counter."""
    global _seed, _offset
    with _offset_lock:
        _seed, _offset = int(seed) & (2 ** 64 - 1), 0


def get_rng_state():
    return _seed, _offset


def set_rng_state(state):
    global _seed, _offset
    with _offset_lock:
        _seed, _offset = int(state[0]), int(state[1])


_offset_lock = threading.Lock()


def _take_offsets(count):
    """Reserve `count` consecutive call offsets (atomic: hooks may run on several host threads)."""
    global _offset
    with _offset_lock:
        o = _offset
        _offset += count
    return o


def _next_offset(rmode):
    if rmode != STOCHASTIC:
        return 0
    return _take_offsets(1)


next_offset = _next_offset
take_offsets = _take_offsets


def cast_(src, dst, fmt, rmode, scale=1.0, seed=None, offset=None, elem_base=0):
    """dst[i] = cast(src[i]); dst may be src.  Both contiguous device tensors of one dtype."""
    _require_device(src)
    _require_device(dst, "output")
    if dst.dtype != src.dtype or dst.numel() != src.numel():
        raise ValueError("fpemu: input/output mismatch")
    if offset is None:
        offset = _next_offset(rmode)
    with _on_device(src):
        _check(lib().fpemu_cast(src.data_ptr(), dst.data_ptr(), src.numel(), _dtype_id(src), fmt, rmode,
                                float(scale), _seed if seed is None else seed, offset, elem_base,
                                _stream(src)))
    return dst


class _TensorEntry(ctypes.Structure):  # fpemu_tensor_t
    _fields_ = [("inp", ctypes.c_void_p), ("out", ctypes.c_void_p), ("n", ctypes.c_uint64)]


def cast_multi_(srcs, fmt, rmode, scale=1.0, dsts=None, seed=None, offset=None):
    """Cast a list of contiguous tensors of one dtype on one device (in place unless `dsts` is given) in
    as few launches as possible.  Tensor i gets call offset `offset + i`: the same bits as a loop of
    cast_() calls, and the call counter advances by len(srcs) exactly as that loop would."""
    srcs = list(srcs)
    dsts = srcs if dsts is None else list(dsts)
    if len(dsts) != len(srcs):
        raise ValueError("fpemu: input/output lists differ in length")
    if not srcs:
        return dsts
    first = srcs[0]
    tab = (_TensorEntry * len(srcs))()
    for i, (a, b) in enumerate(zip(srcs, dsts)):
        _require_device(a)
        _require_device(b, "output")
        if a.dtype != first.dtype or a.device != first.device or b.dtype != a.dtype or b.device != a.device:
            raise ValueError("fpemu: cast_multi_ needs tensors of one dtype on one device")
        if b.numel() != a.numel():
            raise ValueError("fpemu: input/output mismatch")
        tab[i].inp, tab[i].out, tab[i].n = a.data_ptr(), b.data_ptr(), a.numel()
    if offset is None:
        offset = _take_offsets(len(srcs)) if rmode == STOCHASTIC else 0
    with _range("fpemu::cast_multi"), _on_device(first):
        _check(lib().fpemu_cast_multi(ctypes.cast(tab, ctypes.c_void_p), len(srcs), _dtype_id(first), fmt, rmode,
                                      float(scale), _seed if seed is None else seed, offset, _stream(first)))
    return dsts


def quantize_per_tensor_multi_(srcs, fmt, rmode, flt_max, dsts=None, clamp=True, ws=None):
    """Per-tensor amax -> device-side scale -> cast of a list of contiguous tensors of one dtype on one device (in
    place unless `dsts` is given), two launches per 32 tensors.  Tensor i is quantised exactly as
    quantize_per_tensor_() would at call offset `offset + i`; the call counter advances by len(srcs).
    ws: int32 scratch of at least len(srcs) + min(len(srcs), 32) words (default: len + 32 * 256, allocated here);
    afterwards ws[i] holds tensor i's amax bit pattern.  Returns (dsts, ws)."""
    srcs = list(srcs)
    dsts = srcs if dsts is None else list(dsts)
    if len(dsts) != len(srcs):
        raise ValueError("fpemu: input/output lists differ in length")
    if not srcs:
        return dsts, ws
    first = srcs[0]
    tab = (_TensorEntry * len(srcs))()
    for i, (a, b) in enumerate(zip(srcs, dsts)):
        _require_device(a)
        _require_device(b, "output")
        if a.dtype != first.dtype or a.device != first.device or b.dtype != a.dtype or b.device != a.device:
            raise ValueError("fpemu: quantize_per_tensor_multi_ needs tensors of one dtype on one device")
        if b.numel() != a.numel():
            raise ValueError("fpemu: input/output mismatch")
        tab[i].inp, tab[i].out, tab[i].n = a.data_ptr(), b.data_ptr(), a.numel()
    if ws is None:
        ws = torch.empty(len(srcs) + 32 * 256, dtype=torch.int32, device=first.device)
    elif ws.dtype != torch.int32 or ws.device != first.device or not ws.is_contiguous():
        raise ValueError("fpemu: workspace must be a contiguous int32 tensor on the tensors' device")
    offset = _take_offsets(len(srcs)) if rmode == STOCHASTIC else 0
    with _range("fpemu::quantize_per_tensor_multi"), _on_device(first):
        _check(lib().fpemu_quantize_per_tensor_multi(ctypes.cast(tab, ctypes.c_void_p), len(srcs), _dtype_id(first), fmt,
                                                     rmode, float(flt_max), int(bool(clamp)), ws.data_ptr(), ws.numel(),
                                                     _seed, offset, _stream(first)))
    return dsts, ws


def _tensor_table(srcs, dsts, who):
    first = srcs[0]
    tab = (_TensorEntry * len(srcs))()
    for i, (a, b) in enumerate(zip(srcs, dsts)):
        _require_device(a)
        _require_device(b, "output")
        if a.dtype != first.dtype or a.device != first.device or b.dtype != a.dtype or b.device != a.device:
            raise ValueError("fpemu: %s needs tensors of one dtype on one device" % who)
        if b.numel() != a.numel():
            raise ValueError("fpemu: input/output mismatch")
        tab[i].inp, tab[i].out, tab[i].n = a.data_ptr(), b.data_ptr(), a.numel()
    return tab


def _require_words(words, like, count, what="amax words"):
    if (not torch.is_tensor(words) or words.device != like.device or words.dtype != torch.int32
            or not words.is_contiguous() or words.numel() < count):
        raise ValueError("fpemu: %s must be a contiguous int32 tensor of at least %d element(s) on %s"
                         % (what, count, like.device))


def amax_multi_(srcs, words):
    """words[i] (int32, device) <- bit pattern of max|srcs[i]| for a list of contiguous tensors of one dtype on one
    device: the first half of quantize_per_tensor_multi_ for SHARDED tensors -- all-reduce `words` over the process
    group, then cast_multi_scaled_."""
    srcs = list(srcs)
    if not srcs:
        return words
    _require_words(words, srcs[0], len(srcs))
    tab = _tensor_table(srcs, srcs, "amax_multi_")
    with _range("fpemu::amax_multi"), _on_device(srcs[0]):
        _check(lib().fpemu_amax_multi(ctypes.cast(tab, ctypes.c_void_p), len(srcs), _dtype_id(srcs[0]),
                                      words.data_ptr(), _stream(srcs[0])))
    return words


def cast_multi_scaled_(srcs, fmt, rmode, words, flt_max, dsts=None, clamp=True, seed=None, offset=None):
    """Tensor i is cast with the scale derived on the device from words[i] (its amax bit pattern), in place unless
    `dsts` is given; one launch per 32 tensors.  Tensor i draws as call `offset + i`."""
    srcs = list(srcs)
    dsts = srcs if dsts is None else list(dsts)
    if len(dsts) != len(srcs):
        raise ValueError("fpemu: input/output lists differ in length")
    if not srcs:
        return dsts
    _require_words(words, srcs[0], len(srcs))
    tab = _tensor_table(srcs, dsts, "cast_multi_scaled_")
    if offset is None:
        offset = _take_offsets(len(srcs)) if rmode == STOCHASTIC else 0
    with _range("fpemu::cast_multi_scaled"), _on_device(srcs[0]):
        _check(lib().fpemu_cast_multi_scaled(ctypes.cast(tab, ctypes.c_void_p), len(srcs), _dtype_id(srcs[0]), fmt, rmode,
                                             words.data_ptr(), float(flt_max), int(bool(clamp)),
                                             _seed if seed is None else seed, offset, _stream(srcs[0])))
    return dsts


def amax_rows_(src, rows, words):
    """words[r] (int32, device) <- bit pattern of max|row r| of the contiguous buffer viewed as [rows, numel/rows]."""
    _require_device(src)
    n = src.numel()
    if rows <= 0 or n % rows:
        raise ValueError("fpemu: %d elements do not split into %d rows" % (n, rows))
    _require_words(words, src, rows)
    with _range("fpemu::amax_rows"), _on_device(src):
        _check(lib().fpemu_amax_rows(src.data_ptr(), rows, n // rows, _dtype_id(src), words.data_ptr(), _stream(src)))
    return words


def cast_rows_amax_(src, dst, rows, fmt, rmode, words, flt_max, clamp=True, seed=None, offset=None, col_base=0):
    """Row r of the [rows, numel/rows] view is cast with the scale derived on the device from words[r]; row r draws
    as call `offset + r`, its element c as element `col_base + c` (row segments of a sharded tensor)."""
    _require_device(src)
    _require_device(dst, "output")
    n = src.numel()
    if rows <= 0 or n % rows:
        raise ValueError("fpemu: %d elements do not split into %d rows" % (n, rows))
    _require_words(words, src, rows)
    if offset is None:
        offset = _take_offsets(rows) if rmode == STOCHASTIC else 0
    with _range("fpemu::cast_rows_amax"), _on_device(src):
        _check(lib().fpemu_cast_rows_amax(src.data_ptr(), dst.data_ptr(), rows, n // rows, _dtype_id(src), fmt, rmode,
                                          words.data_ptr(), float(flt_max), int(bool(clamp)),
                                          _seed if seed is None else seed, offset, int(col_base), _stream(src)))
    return dst


def amax_(src, amax_bits):
    """amax_bits (int32[1], same device, zeroed by the caller) <- max(amax_bits, bits(max|src|))."""
    _require_device(src)
    with _on_device(src):
        _check(lib().fpemu_amax(src.data_ptr(), src.numel(), _dtype_id(src), amax_bits.data_ptr(),
                                _stream(src)))
    return amax_bits


def cast_scaled_(src, dst, fmt, rmode, amax_bits, flt_max, clamp=True, seed=None, offset=None,
                 elem_base=0):
    _require_device(src)
    _require_device(dst, "output")
    if offset is None:
        offset = _next_offset(rmode)
    with _on_device(src):
        _check(lib().fpemu_cast_scaled(src.data_ptr(), dst.data_ptr(), src.numel(), _dtype_id(src), fmt,
                                       rmode, amax_bits.data_ptr(), float(flt_max), int(bool(clamp)),
                                       _seed if seed is None else seed, offset, elem_base, _stream(src)))
    return dst


def _require_scales(scales, like, count):
    if (not torch.is_tensor(scales) or scales.device != like.device or scales.dtype != torch.float32
            or not scales.is_contiguous() or scales.numel() < count):
        raise ValueError("fpemu: scales must be a contiguous float32 tensor of at least %d element(s) on %s"
                         % (count, like.device))


def cast_device_scale_(src, dst, fmt, rmode, scale, seed=None, offset=None, elem_base=0):
    """Cast with the scale read on the device from scale[0] (float32 tensor), no host sync."""
    _require_device(src)
    _require_device(dst, "output")
    _require_scales(scale, src, 1)
    if offset is None:
        offset = _next_offset(rmode)
    with _range("fpemu::cast_device_scale"), _on_device(src):
        _check(lib().fpemu_cast_device_scale(src.data_ptr(), dst.data_ptr(), src.numel(), _dtype_id(src), fmt, rmode,
                                             scale.data_ptr(), _seed if seed is None else seed, offset, elem_base,
                                             _stream(src)))
    return dst


def cast_rows_scaled_(src, dst, rows, fmt, rmode, scales, offset=None):
    """Cast of a contiguous buffer viewed as [rows, numel/rows] with scales[r] (float32, device) for row r, one
    launch; equals a loop of per-row casts with consecutive call offsets."""
    _require_device(src)
    _require_device(dst, "output")
    n = src.numel()
    if rows <= 0 or n % rows:
        raise ValueError("fpemu: %d elements do not split into %d rows" % (n, rows))
    _require_scales(scales, src, rows)
    off = offset if offset is not None else (_take_offsets(rows) if rmode == STOCHASTIC else 0)
    with _range("fpemu::cast_rows_scaled"), _on_device(src):
        _check(lib().fpemu_cast_rows_scaled(src.data_ptr(), dst.data_ptr(), rows, n // rows, _dtype_id(src), fmt,
                                            rmode, scales.data_ptr(), _seed, off, _stream(src)))
    return dst


_WS_WORDS = 1 + 256
_ws_cache = {}
_tls = threading.local()


class use_workspace(object):
    """Context: per-tensor quantisation on this host thread uses `ws` (int32, >= 257 words, allocated by the
    caller) instead of the cached per-stream scratch.  mpemu.graphs.GraphedForward captures under it, so every
    graph owns the scratch whose address it bakes in."""

    def __init__(self, ws):
        self.ws = ws

    def __enter__(self):
        self.prev = getattr(_tls, "ws", None)
        _tls.ws = self.ws
        return self.ws

    def __exit__(self, *a):
        _tls.ws = self.prev
        return False


def _stream_workspace(t, stream):
    """Scratch words for the per-tensor path, one buffer per (device, stream): calls on one stream are
    ordered, calls on different streams must not share scratch.  Never cached while the stream is being
    captured into a graph: a buffer allocated there belongs to that graph's private pool, and a later capture
    (same capture-stream handle) must not bake its address in -- such calls get a buffer of their own."""
    ws = getattr(_tls, "ws", None)
    if ws is not None and ws.device == t.device:
        return ws
    key = (t.device.index, stream)
    ws = _ws_cache.get(key)
    if ws is None:
        ws = torch.empty(_WS_WORDS, dtype=torch.int32, device=t.device)
        if not torch.cuda.is_current_stream_capturing():
            _ws_cache[key] = ws
    return ws


def stream_workspace(t):
    """The scratch buffer (int32[257]) of torch's current stream on t's device."""
    with _on_device(t):
        return _stream_workspace(t, _stream(t))


def quantize_per_tensor_(src, dst, fmt, rmode, flt_max, amax_ws=None, clamp=True):
    """Fused amax -> device-side scale -> cast (qutils.py:282-287), no host sync.  amax_ws: int32 tensor
    whose word 0 receives the amax bit pattern (1 word, or more to serve as scratch); default: a cached
    per-stream scratch buffer."""
    _require_device(src)
    _require_device(dst, "output")
    with _range("fpemu::quantize_per_tensor"), _on_device(src):
        stream = _stream(src)
        if amax_ws is None:
            amax_ws = _stream_workspace(src, stream)
        _check(lib().fpemu_quantize_per_tensor_ws(src.data_ptr(), dst.data_ptr(), src.numel(), _dtype_id(src),
                                                  fmt, rmode, float(flt_max), int(bool(clamp)),
                                                  amax_ws.data_ptr(), amax_ws.numel(), _seed,
                                                  _next_offset(rmode), stream))
    return dst


def quantize_rows_(src, dst, rows, fmt, rmode, flt_max, clamp=True, offset=None):
    """Per-row amax -> scale -> cast of a contiguous buffer viewed as [rows, numel/rows], one launch
    (the per-channel / fine-grained policies of qutils.py:289-337)."""
    _require_device(src)
    _require_device(dst, "output")
    n = src.numel()
    if rows <= 0 or n % rows:
        raise ValueError("fpemu: %d elements do not split into %d rows" % (n, rows))
    # one call offset per row, as a loop of per-row casts would consume
    off = offset if offset is not None else (_take_offsets(rows) if rmode == STOCHASTIC else 0)
    with _range("fpemu::quantize_rows"), _on_device(src):
        _check(lib().fpemu_quantize_rows(src.data_ptr(), dst.data_ptr(), rows, n // rows, _dtype_id(src),
                                         fmt, rmode, float(flt_max), int(bool(clamp)), _seed, off,
                                         _stream(src)))
    return dst


_BLOCKNORM_FMTS = (0, 1, 2, 7, 8)  # E5M2, E4M3, E3M4, E4M3_IEEE, FP4: reference kernels with a block_norm branch


def quantize_blocks_(src, dst, block_size, fmt, rmode):
    """Block-normalised cast (forward(..., block_norm=True, block_size) of the reference): per block of
    block_size consecutive elements the scale comes from the block's max |x|; one launch."""
    _require_device(src)
    _require_device(dst, "output")
    n = src.numel()
    if block_size <= 0 or n % block_size:
        raise ValueError("fpemu: %d elements do not split into blocks of %d" % (n, block_size))
    off = _take_offsets(n // block_size) if rmode == STOCHASTIC else 0
    with _range("fpemu::quantize_blocks"), _on_device(src):
        _check(lib().fpemu_quantize_blocks(src.data_ptr(), dst.data_ptr(), n, block_size, _dtype_id(src), fmt,
                                           rmode, _seed, off, _stream(src)))
    return dst


GEMM_SEPARATE, GEMM_FMA = 0, 1


# ---- known output maxima ---------------------------------------------------------------------
# A c-model op whose epilogue folded max |output| into a device word leaves (output, word) here; the per-tensor
# quantisation of that very tensor right afterwards (quantize_module_output, reference qutils.py:607-609) then skips
# its amax pass.  ONE slot, holding the tensor itself: while it is held its address cannot be handed to another
# tensor, and any in-place change since bumps _version -- a stale entry can never match.  Per host thread (hooks may
# run on several): a producer on one thread never clobbers or feeds another thread's consumer.
class _Known(threading.local):
    slot = None


_known_tls = _Known()


def note_amax(t, word):
    """`word` (int32[1] on t's device) holds bits(max |t|) in stream order; t is as the producer left it."""
    _known_tls.slot = (t, t.data_ptr(), t.numel(), t._version, word)


def has_known_amax():
    return _known_tls.slot is not None


def take_amax(t):
    """The amax word of t if the last c-model op on this thread produced exactly this tensor (same storage, size,
    dtype, version), else None.  Either way the slot is emptied."""
    k, _known_tls.slot = _known_tls.slot, None
    if (k is not None and k[1] == t.data_ptr() and k[2] == t.numel() and k[3] == t._version
            and k[0].dtype == t.dtype and k[0].device == t.device and t.is_contiguous()):
        return k[4]
    return None


def forget_amax():
    _known_tls.slot = None


def gemm_(C, A, B, alpha=1.0, a_trans=False, b_trans=False, variant=GEMM_FMA, amax=None):
    """C += alpha * op(A) @ op(B) with the cmodel's k-ascending fp32 accumulation chain; the call
    simple_gemm_dev.gemm(C, A, B, alpha, a_trans, b_trans) of the reference
    (mpemu/cmodel/simple/simple_gemm.cpp:45-67).  A is stored [K, M] when a_trans, B [N, K] when
    b_trans.  variant: GEMM_FMA (fmaf chain on MFMA) or GEMM_SEPARATE (multiply and add rounded
    separately, VALU).  amax: int32[1] on C's device (zeroed by the caller): receives
    max(amax, bits(max |C|)) of the final C from the kernel's epilogue (fpemu_gemm_f32_amax)."""
    for t, what in ((C, "C"), (A, "A"), (B, "B")):
        _require_device(t, what)
        if t.dtype != torch.float32 or t.dim() != 2:
            raise TypeError("fpemu gemm: %s must be a 2-D float32 tensor" % what)
    M, N = C.shape
    K = A.shape[0] if a_trans else A.shape[1]
    if (A.shape[1] if a_trans else A.shape[0]) != M or (B.shape[0] if b_trans else B.shape[1]) != N or \
            (B.shape[1] if b_trans else B.shape[0]) != K:
        raise ValueError("fpemu gemm: shapes do not match: C %s, A %s (trans=%s), B %s (trans=%s)" % (
            tuple(C.shape), tuple(A.shape), a_trans, tuple(B.shape), b_trans))
    with _range("cmodel::simple_gemm"), _on_device(C):
        if amax is None:
            _check(lib().fpemu_gemm_f32(C.data_ptr(), A.data_ptr(), B.data_ptr(), M, N, K, A.shape[1], B.shape[1],
                                        C.shape[1], float(alpha), int(bool(a_trans)), int(bool(b_trans)),
                                        int(variant), _stream(C)))
        else:
            _require_amax_word(amax, C)
            _check(lib().fpemu_gemm_f32_amax(C.data_ptr(), A.data_ptr(), B.data_ptr(), M, N, K, A.shape[1], B.shape[1],
                                             C.shape[1], float(alpha), int(bool(a_trans)), int(bool(b_trans)),
                                             int(variant), amax.data_ptr(), _stream(C)))
    return C


def _require_amax_word(word, like):
    if (not torch.is_tensor(word) or word.device != like.device or word.dtype != torch.int32 or word.numel() < 1
            or not word.is_contiguous()):
        raise ValueError("fpemu: amax must be a contiguous int32 tensor of at least one word on %s" % (like.device,))


CONV_FP, CONV_BP, CONV_WU = 0, 1, 2


def _conv_ws(kind, dims, like):
    nbytes = lib().fpemu_conv2d_workspace_bytes(kind, *dims)
    if nbytes == 0:
        _check(-1 if lib().fpemu_last_error() else 0)
    return torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=like.device), nbytes


def _conv_check(ts):
    for t, what in ts:
        _require_device(t, what)
        if t.dtype != torch.float32:
            raise TypeError("fpemu conv2d: %s must be float32" % what)


def conv2d_fp_(output, input, weight, bias, stride, padding, variant=GEMM_FMA, amax=None):
    """simple_conv2d_dev.conv2d_fp(output, input, weight, bias, stride, padding, 1, 1) of the reference
    (cmodel/simple/simple_conv2d.cpp:44-69) on the GPU; output [N,K,OH,OW] is overwritten.  amax: as gemm_."""
    _conv_check([(output, "output"), (input, "input"), (weight, "weight")] + ([(bias, "bias")] if bias is not None else []))
    N, C, H, W = input.shape
    K, C1, R, S = weight.shape
    dims = (N, C, H, W, K, R, S, int(stride), int(padding))
    OH, OW = (H + 2 * padding - R) // stride + 1, (W + 2 * padding - S) // stride + 1
    if C1 != C or tuple(output.shape) != (N, K, OH, OW) or (bias is not None and bias.numel() != K):
        raise ValueError("fpemu conv2d_fp: shapes do not match")
    with _range("cmodel::simple_conv2d_fp"), _on_device(input):
        ws, nbytes = _conv_ws(CONV_FP, dims, input)
        if amax is None:
            _check(lib().fpemu_conv2d_fp_f32(output.data_ptr(), input.data_ptr(), weight.data_ptr(),
                                             None if bias is None else bias.data_ptr(), *dims, int(variant),
                                             ws.data_ptr(), nbytes, _stream(input)))
        else:
            _require_amax_word(amax, output)
            _check(lib().fpemu_conv2d_fp_f32_amax(output.data_ptr(), input.data_ptr(), weight.data_ptr(),
                                                  None if bias is None else bias.data_ptr(), *dims, int(variant),
                                                  ws.data_ptr(), nbytes, amax.data_ptr(), _stream(input)))
    return output


def conv2d_bp_(grad_input, grad_output, weight, stride, padding, variant=GEMM_FMA):
    """conv2d_bp of the reference (simple_conv2d.cpp:71-95): grad_input [N,C,H,W] is overwritten."""
    _conv_check([(grad_input, "grad_input"), (grad_output, "grad_output"), (weight, "weight")])
    N, C, H, W = grad_input.shape
    K, C1, R, S = weight.shape
    dims = (N, C, H, W, K, R, S, int(stride), int(padding))
    OH, OW = (H + 2 * padding - R) // stride + 1, (W + 2 * padding - S) // stride + 1
    if C1 != C or tuple(grad_output.shape) != (N, K, OH, OW):
        raise ValueError("fpemu conv2d_bp: shapes do not match")
    with _range("cmodel::simple_conv2d_bp"), _on_device(grad_input):
        ws, nbytes = _conv_ws(CONV_BP, dims, grad_input)
        _check(lib().fpemu_conv2d_bp_f32(grad_input.data_ptr(), grad_output.data_ptr(), weight.data_ptr(), *dims,
                                         int(variant), ws.data_ptr(), nbytes, _stream(grad_input)))
    return grad_input


def conv2d_wu_(grad_weight, grad_output, input, stride, padding, variant=GEMM_FMA):
    """conv2d_wu of the reference (simple_conv2d.cpp:97-121): grad_weight [K,C,R,S] is overwritten."""
    _conv_check([(grad_weight, "grad_weight"), (grad_output, "grad_output"), (input, "input")])
    N, C, H, W = input.shape
    K, C1, R, S = grad_weight.shape
    dims = (N, C, H, W, K, R, S, int(stride), int(padding))
    OH, OW = (H + 2 * padding - R) // stride + 1, (W + 2 * padding - S) // stride + 1
    if C1 != C or tuple(grad_output.shape) != (N, K, OH, OW):
        raise ValueError("fpemu conv2d_wu: shapes do not match")
    with _range("cmodel::simple_conv2d_wu"), _on_device(input):
        ws, nbytes = _conv_ws(CONV_WU, dims, input)
        _check(lib().fpemu_conv2d_wu_f32(grad_weight.data_ptr(), grad_output.data_ptr(), input.data_ptr(), *dims,
                                         int(variant), ws.data_ptr(), nbytes, _stream(input)))
    return grad_weight


def forward(input, mode, size, inplace, scale, block_norm=False, block_size=1):
    """Same call as the reference's pybind `fpemu_hip.forward` (hip/fpemu_impl.cpp:27-37):
    returns a 1-element list holding the new tensor, or `input` itself when inplace."""
    _require_device(input)
    if int(size) != input.numel():
        raise ValueError("fpemu.forward: size (%d) != input.numel() (%d)" % (size, input.numel()))
    fmt, rmode = parse_mode(mode)
    out = input if inplace else torch.empty_like(input)
    n, bs = input.numel(), int(block_size)
    if block_norm and bs > 0 and n > 0 and (bs == n or n % bs == 0) and fmt in _BLOCKNORM_FMTS:
        return [quantize_blocks_(input, out, bs, fmt, rmode)]  # cpp:2020-2025: otherwise demoted
    with _range("fpemu::forward"), _on_device(input):
        _check(lib().fpemu_forward(input.data_ptr(), out.data_ptr(), _dtype_id(input), str(mode).encode(),
                                   input.numel(), float(scale), 0, 1,
                                   _seed, _next_offset(rmode), _stream(input)))
    return [out]


backward = forward  # hip/fpemu_impl.cpp:39-49: the same function applied to a gradient
