"""Test-only stand-in for the HIP binding: same call surface (parse_mode / cast_ / amax_ /
cast_scaled_) on CPU tensors, implemented with the oracle.  Lets the multi-process collective
logic of mpemu.sharded run over gloo in a container without GPUs.  Never imported by the product."""
import numpy as np
import torch

import orc

_FMT_NAMES = {v: k for k, v in orc.FMT.items()}
_RM_NAMES = {v: k for k, v in orc.RMODE.items()}
_MODE_PREFIX = {"BF16": "BFLOAT16", "FP16": "FLOAT16", "FP16_DAZ": "FLOAT16_DAZ"}


def parse_mode(mode):
    return orc.parse_mode(mode)


def _mode(fmt, rmode):
    f = _FMT_NAMES[fmt]
    return "%s_%s" % (_MODE_PREFIX.get(f, f), _RM_NAMES[rmode])


def cast_(src, dst, fmt, rmode, scale=1.0, seed=None, offset=None, elem_base=0):
    y = orc.cast(src.numpy(), _mode(fmt, rmode), scale=scale, seed=seed or 0, offset=offset or 0,
                 elem_base=elem_base)
    dst.copy_(torch.from_numpy(y).view_as(dst))
    return dst


def amax_(src, amax_bits):
    b = int(np.array([orc.amax(src.numpy())], dtype=np.float32).view(np.uint32)[0])
    amax_bits[0] = max(int(amax_bits[0]), b)
    return amax_bits


def cast_scaled_(src, dst, fmt, rmode, amax_bits, flt_max, clamp=True, seed=None, offset=None, elem_base=0):
    vmax = np.array([int(amax_bits[0])], dtype=np.int64).astype(np.uint32).view(np.float32)[0]
    s = orc.scale_from_amax(vmax, flt_max, clamp=clamp)
    return cast_(src, dst, fmt, rmode, scale=s, seed=seed, offset=offset, elem_base=elem_base)


# ---- what mpemu.qutils' process-group path uses besides the calls above --------------------------------
STOCHASTIC = 2
_offset = 0
_ws = torch.zeros(257, dtype=torch.int32)


def next_offset(rmode):
    global _offset
    if rmode != STOCHASTIC:
        return 0
    _offset += 1
    return _offset - 1


def set_rng_state(state):
    global _offset
    _offset = int(state[1])


def stream_workspace(t):
    return _ws


def take_offsets(count):
    global _offset
    o = _offset
    _offset += int(count)
    return o


def get_rng_state():
    return 0, _offset


FMT_BF16, FMT_FP16, FMT_FP16_DAZ = orc.FMT["BF16"], orc.FMT["FP16"], orc.FMT["FP16_DAZ"]


def _word(a):
    return int(np.array([orc.amax(a)], dtype=np.float32).view(np.uint32)[0])


def _vmax(word):
    return np.array([int(word)], dtype=np.int64).astype(np.uint32).view(np.float32)[0]


def amax_rows_(src, rows, words):
    a = src.numpy().reshape(rows, -1)
    for r in range(rows):
        words[r] = np.int32(np.uint32(_word(a[r])).view(np.int32)) if a.shape[1] else 0
    return words


def cast_rows_amax_(src, dst, rows, fmt, rmode, words, flt_max, clamp=True, seed=None, offset=None, col_base=0):
    a = src.numpy().reshape(rows, -1)
    d = dst.view(rows, -1)
    if offset is None:
        offset = take_offsets(rows) if rmode == STOCHASTIC else 0
    for r in range(rows):
        s = orc.scale_from_amax(_vmax(words[r]), flt_max, clamp=clamp)
        y = orc.cast(a[r], _mode(fmt, rmode), scale=s, seed=seed or 0, offset=offset + r, elem_base=col_base)
        d[r].copy_(torch.from_numpy(y))
    return dst


def quantize_rows_(src, dst, rows, fmt, rmode, flt_max, clamp=True, offset=None):
    words = torch.zeros(rows, dtype=torch.int32)
    amax_rows_(src, rows, words)
    return cast_rows_amax_(src, dst, rows, fmt, rmode, words, flt_max, clamp=clamp, offset=offset)


def amax_multi_(srcs, words):
    for i, t in enumerate(srcs):
        words[i] = np.int32(np.uint32(_word(t.numpy())).view(np.int32)) if t.numel() else 0
    return words


def cast_multi_scaled_(srcs, fmt, rmode, words, flt_max, dsts=None, clamp=True, seed=None, offset=None):
    dsts = srcs if dsts is None else dsts
    if offset is None:
        offset = take_offsets(len(srcs)) if rmode == STOCHASTIC else 0
    for i, (a, b) in enumerate(zip(srcs, dsts)):
        cast_scaled_(a, b, fmt, rmode, words[i:i + 1], flt_max, clamp=clamp, seed=seed, offset=offset + i)
    return dsts


def ext():
    return None


def tracing():
    return False
