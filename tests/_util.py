"""Shared helpers for the test-suite (test infrastructure)."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

FLT_MAX = {"E5M2": 57344.0, "E4M3": 448.0, "E3M4": 30.0}
FLT_MIN = {"E5M2": 1.5258789E-05, "E4M3": 1.9531250E-03, "E3M4": 1.5625000E-02}  # reference qutils.py get_flt_min

DET_MODES = ["E5M2_RNE", "E4M3_RNE", "E3M4_RNE", "E5M2_RTZ", "E5M2_RNAZ", "E5M2_RNTZ", "E5M2_RPINF",
             "E5M2_RNINF", "E5M2_DAZ_RNE", "E5M2_DAZ_RNAZ", "E5M2_DAZ_RNTZ", "BFLOAT16_RNE",
             "FLOAT16_RNE", "FLOAT16_DAZ_RNE"]
EXT_MODES = ["E4M3_RTZ", "E3M4_RTZ", "BFLOAT16_RTZ"]  # not defined by the reference: parity unpinned
SR_MODES = ["E5M2_STOCHASTIC", "E4M3_STOCHASTIC", "E3M4_STOCHASTIC", "BFLOAT16_STOCHASTIC",
            "FLOAT16_STOCHASTIC", "E5M2_DAZ_STOCHASTIC"]
# pinned by tests/golden/ext2.npz (E4M3_IEEE: the reference's AVX-512 routines)
F4_MODES = ["E4M3_IEEE_RNE", "FP4_NEAREST"]
F4_SR_MODES = ["E4M3_IEEE_STOCHASTIC"]
BLOCKNORM_MODES = ["E5M2_RNE", "E5M2_RTZ", "E5M2_RNAZ", "E4M3_RNE", "E3M4_RNE", "E4M3_IEEE_RNE", "FP4_NEAREST"]
ALL_MODES = DET_MODES + EXT_MODES + SR_MODES + F4_MODES + F4_SR_MODES


def npz(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def jload(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def assert_bits_equal(got, want, what="", nan_payload_ok=False):
    """Bit-for-bit equality of two fp32 arrays.  nan_payload_ok: positions where BOTH sides are
    NaN count as equal whatever sign/payload the NaN carries (only for NaNs that arithmetic
    generates, e.g. 0 * inf, whose bit pattern is an ISA convention: x86 0xFFC00000, gfx950
    0x7FC00000)."""
    g, w = bits(got).ravel(), bits(want).ravel()
    assert g.shape == w.shape, (what, g.shape, w.shape)
    neq = g != w
    if nan_payload_ok:
        both_nan = ((g & 0x7FFFFFFF) > 0x7F800000) & ((w & 0x7FFFFFFF) > 0x7F800000)
        neq &= ~both_nan
    bad = np.nonzero(neq)[0]
    if bad.size:
        i = bad[:8]
        raise AssertionError("%s: %d / %d words differ; first at %s got %s want %s" % (
            what, bad.size, g.size, i.tolist(), [hex(v) for v in g[i]], [hex(v) for v in w[i]]))


import functools


@functools.lru_cache(maxsize=4)
def _mixed_cached(n, seed):
    return _mixed_inputs(n, seed)


def mixed_inputs(n, seed=1234):
    """Cached for the big sizes; returns a private copy."""
    return _mixed_cached(n, seed).copy()


def _mixed_inputs(n, seed=1234):
    """BASELINE config-2 style mix: 90 % log-normal magnitudes, 5 % exact zeros / format maxima /
    ties, 5 % raw random bit patterns (NaN / Inf / denormals)."""
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal(n) * np.exp(rng.standard_normal(n) * 4.0)).astype(np.float32)
    k = n // 20
    special = np.array([0.0, -0.0, 448.0, -448.0, 57344.0, -57344.0, 30.0, -30.0, 65504.0, 61440.0,
                        1.0625, 1.125, 1.25, 0.001953125, 2.0 ** -10, 2.0 ** -16, 2.0 ** -17,
                        1.5 * 2.0 ** -16, np.inf, -np.inf, 464.0, 480.0, 31.0, 30.5],
                       dtype=np.float32)
    idx = rng.choice(n, size=2 * k, replace=False)
    x[idx[:k]] = special[rng.integers(0, special.size, size=k)]
    raw = rng.integers(0, 2 ** 32, size=k, dtype=np.uint64).astype(np.uint32)
    x.view(np.uint32)[idx[k:]] = raw
    return x
