"""Stochastic-rounding acceptance tests: unbiasedness, histogram (chi-square over all
residues), independence (lag-1..63 autocorrelation of the round-up indicator) and determinism.  The CPU
half runs on the oracle's Philox stream; the GPU half on the kernels (which the parity tests pin to that
stream bit for bit) plus grid-size independence."""
import math

import numpy as np
import pytest

import orc
from _util import bits

N = 1 << 20
CASES = [("E5M2_STOCHASTIC", 8), ("E4M3_STOCHASTIC", 7), ("E3M4_STOCHASTIC", 6), ("E4M3_IEEE_STOCHASTIC", 7),
         ("E5M2_DAZ_STOCHASTIC", 8)]


def _const_input(k, r):
    """fp16 pattern with exponent 0 (1 <= x < 2), kept-mantissa LSB set, dropped-field residue r."""
    base = 0x3C00 | (1 << k)
    h = np.array([base | r, base, base + (1 << k)], dtype=np.uint16).view(np.float16).astype(np.float32)
    return h[0], h[1], h[2]


def _check_unbiased(cast, mode, k):
    for r in (1, (1 << k) // 4 + 1, (1 << k) // 2, (1 << k) - 1):
        x, lo, hi = _const_input(k, r)
        out = cast(np.full(N, x, dtype=np.float32), mode, 17, r)
        up = out == hi
        assert np.all(up | (out == lo))
        p = r / float(1 << k)
        sigma = math.sqrt(N * p * (1 - p))
        assert abs(int(up.sum()) - N * p) < 5 * sigma, (mode, r, int(up.sum()), N * p)
        # mean(out) within the matching bound of the (fp16-exact) input
        assert abs(float(out.astype(np.float64).mean()) - float(x)) < 5 * sigma / N * float(hi - lo)
    x, lo, hi = _const_input(k, 0)
    assert np.all(cast(np.full(4096, x, dtype=np.float32), mode, 17, 0) == x)   # p = 0: never moves


def _check_histogram(cast, mode, k):
    """All 2^k residues at one exponent, 16384 draws each: chi-square of round-up counts vs r / 2^k."""
    n_each = 16384
    res = np.arange(1, 1 << k)
    base = 0x3C00 | (1 << k)
    x = np.repeat((base | res).astype(np.uint16).view(np.float16).astype(np.float32), n_each)
    hi = np.array([base + (1 << k)], dtype=np.uint16).view(np.float16).astype(np.float32)[0]
    up = (cast(x, mode, 99, 5) == hi).reshape(len(res), n_each).sum(axis=1)
    p = res / float(1 << k)
    chi2 = float((((up - n_each * p) ** 2) / (n_each * p * (1 - p))).sum())
    dof = len(res)
    assert chi2 < dof + 6 * math.sqrt(2 * dof), (mode, chi2, dof)


def _check_independence(cast, mode, k):
    """Round-up indicators of neighbouring elements are uncorrelated at every lag up to 63 and at the
    vector / lane / superblock strides of the noise layout (reusing random bytes across lanes would show)."""
    x, lo, hi = _const_input(k, (1 << k) // 2)            # p = 1/2: maximal variance
    up = (cast(np.full(N, x, dtype=np.float32), mode, 5, 1) == hi).astype(np.float64) - 0.5
    bound = 6.0 / math.sqrt(N)                              # each lag's estimate ~ N(0, 1/N) on +-0.5 -> x4
    for lag in list(range(1, 64)) + [64, 128, 256, 1024, 4096]:
        c = float((up[:-lag] * up[lag:]).mean()) * 4.0
        assert abs(c) < bound, (mode, lag, c)


def _oracle_cast(x, mode, seed, offset):
    return orc.cast(x, mode, seed=seed, offset=offset)


@pytest.mark.parametrize("mode,k", CASES)
def test_oracle_stream_unbiased_histogram_independent(mode, k):
    _check_unbiased(_oracle_cast, mode, k)
    _check_histogram(_oracle_cast, mode, k)
    _check_independence(_oracle_cast, mode, k)


def test_oracle_bf16_and_fp16_stochastic_unbiased():
    for mode, kbits, base in (("BFLOAT16_STOCHASTIC", 16, 0x3F810000), ("FLOAT16_STOCHASTIC", 13, 0x3F802000)):
        for r in (1, (1 << kbits) // 3, (1 << kbits) - 1):
            x = np.full(N, np.array([base | r], dtype=np.uint32).view(np.float32)[0], dtype=np.float32)
            out = bits(orc.cast(x, mode, seed=3, offset=r))
            up = out == (base + (1 << kbits))
            assert np.all(up | (out == base))
            p = r / float(1 << kbits)
            assert abs(int(up.sum()) - N * p) < 5 * math.sqrt(max(N * p * (1 - p), 1.0)), (mode, r)


@pytest.mark.gpu
@pytest.mark.parametrize("mode,k", CASES)
def test_gpu_unbiased_histogram_independent_and_grid_invariant(mode, k):
    import torch

    from mpemu.pytquant.hip import _binding as hip

    fmt, rm = hip.parse_mode(mode)

    def gpu_cast(x, mode_, seed, offset):
        t = torch.from_numpy(np.ascontiguousarray(x)).to("cuda:0")
        o = torch.empty_like(t)
        hip.cast_(t, o, fmt, rm, seed=seed, offset=offset)
        return o.cpu().numpy()

    _check_unbiased(gpu_cast, mode, k)
    _check_histogram(gpu_cast, mode, k)
    _check_independence(gpu_cast, mode, k)
    # determinism: same (seed, offset) -> same bits whatever the launch geometry; other offset -> other draws
    x, _, _ = _const_input(k, 37 % (1 << k) or 1)
    xs = np.full((1 << 18) + 333, x, dtype=np.float32)
    ref = gpu_cast(xs, mode, 8, 2)
    try:
        for cap in (1, 7, 64, 1000):
            hip.lib().fpemu_test_set_max_blocks_(cap)
            assert np.array_equal(bits(gpu_cast(xs, mode, 8, 2)), bits(ref)), (mode, cap)
    finally:
        hip.lib().fpemu_test_set_max_blocks_(0)
    assert not np.array_equal(bits(gpu_cast(xs, mode, 8, 3)), bits(ref))
    assert np.array_equal(bits(ref), bits(orc.cast(xs, mode, seed=8, offset=2)))
