"""Randomised parity: the HIP cast against the oracle over random (mode, storage, size, scale, alignment,
grid cap, shard position) draws -- every path a caller can reach through fpemu_cast: 16-byte and per-lane
paths, capped (looping) grids, non-zero elem_base (shards), in-place, fp32 / fp16 / bf16 storage, deterministic
and stochastic modes (stochastic: the oracle reproduces the Philox stream, so the compare is bit-exact)."""
import numpy as np
import pytest
import torch

import orc
from _util import ALL_MODES, assert_bits_equal, mixed_inputs

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
CASES = 400


@pytest.fixture(scope="module")
def hip():
    from mpemu.pytquant.hip import _binding as b

    b.lib()
    yield b
    b.lib().fpemu_test_set_max_blocks_(0)


def test_random_cases_vs_oracle(hip):
    rng = np.random.default_rng(20261110)
    for case in range(CASES):
        mode = ALL_MODES[int(rng.integers(len(ALL_MODES)))]
        fmt, rm = hip.parse_mode(mode)
        storage = ("f32", "f32", "f16", "bf16")[int(rng.integers(4))]
        n = int(rng.choice([1, 5, 63, 255, 256, 1023, 1024, 1025, 2047, 4096, 10007, 65536, 250000, (1 << 20) + 3]))
        scale = float(rng.choice([1.0, 1.0, 0.37, 3.0, 116.894, 6.55e4]))
        shift = int(rng.integers(0, 4))                     # leading elements dropped: misaligned views
        cap = int(rng.choice([0, 0, 1, 7, 64]))             # fpemu_test_set_max_blocks_: looping grids
        sb = 2048 if storage != "f32" else 1024             # stochastic shards start on a superblock of the layout
        elem_base = int(rng.integers(0, 1 << 20)) * sb
        inplace = bool(rng.integers(2))
        seed, offset = int(rng.integers(1 << 40)), int(rng.integers(1 << 20))
        what = "case %d: %s %s n=%d scale=%r shift=%d cap=%d base=%d inplace=%s" % (
            case, mode, storage, n, scale, shift, cap, elem_base, inplace)
        x = mixed_inputs(n + shift, seed=case)
        hip.lib().fpemu_test_set_max_blocks_(cap)
        try:
            if storage == "f32":
                t = torch.from_numpy(x).to(DEV)[shift:]
                o = t if inplace else torch.empty(n + shift, dtype=torch.float32, device=DEV)[shift:]
                hip.cast_(t, o, fmt, rm, scale=scale, seed=seed, offset=offset, elem_base=elem_base)
                torch.cuda.synchronize()
                want = orc.cast(x[shift:], mode, scale=scale, seed=seed, offset=offset, elem_base=elem_base)
                assert_bits_equal(o.cpu().numpy(), want, what)
            else:
                dt = torch.float16 if storage == "f16" else torch.bfloat16
                with np.errstate(over="ignore", invalid="ignore"):
                    t = torch.from_numpy(x).to(DEV).to(dt)[shift:]
                o = t.clone() if inplace else torch.empty(n + shift, dtype=dt, device=DEV)[shift:]
                src = o if inplace else t
                bits_in = t.view(torch.int16).cpu().numpy().view(np.uint16)
                hip.cast_(src, o, fmt, rm, scale=scale, seed=seed, offset=offset, elem_base=elem_base)
                torch.cuda.synchronize()
                want = orc.cast16(bits_in, storage == "bf16", mode, scale=scale, seed=seed, offset=offset,
                                  elem_base=elem_base)
                got = o.view(torch.int16).cpu().numpy().view(np.uint16)
                # NaN payloads of 16-bit storage are the hardware convert's: compare NaN-ness there
                exp_mask = 0x7C00 if storage == "f16" else 0x7F80
                man_mask = 0x03FF if storage == "f16" else 0x007F
                nan = ((want & exp_mask) == exp_mask) & ((want & man_mask) != 0)
                gnan = ((got & exp_mask) == exp_mask) & ((got & man_mask) != 0)
                assert np.array_equal(nan, gnan), what
                assert np.array_equal(got[~nan], want[~nan]), what
        finally:
            hip.lib().fpemu_test_set_max_blocks_(0)


def test_random_rows_and_blocks_vs_oracle(hip):
    """Per-row (per-channel, clamped or not) and block-norm quantisation over random shapes -- every team shape of
    the rows kernels (sub-vector, lane groups of 1..256 lanes with 1..8 held vectors, 1024-thread groups with 4..16, two-pass wave / block
    teams, unaligned rows) -- against the oracle's loop of per-row casts."""
    from _util import FLT_MAX

    rng = np.random.default_rng(7)
    modes = ["E4M3_RNE", "E5M2_RNE", "E3M4_RNE", "E4M3_STOCHASTIC", "E5M2_STOCHASTIC", "E4M3_RTZ", "E5M2_RNAZ"]
    col_choices = [1, 2, 3, 4, 7, 8, 16, 20, 64, 100, 256, 260, 768, 1000, 1024, 1028, 4096, 8192, 8196, 11008, 12288, 16384, 20000, 32768, 50000, 65536, 65540]
    for case in range(120):
        mode = modes[int(rng.integers(len(modes)))]
        fmt, rm = hip.parse_mode(mode)
        cols = int(rng.choice(col_choices))
        rows = int(rng.integers(1, max(2, min(600, (1 << 21) // cols))))
        kind = ("clamp", "noclamp", "blocknorm")[int(rng.integers(3))]
        if kind == "blocknorm" and mode in ("E4M3_RTZ",):
            kind = "clamp"
        x = (rng.standard_normal((rows, cols)) * np.exp(rng.standard_normal((rows, 1)) * 5)).astype(np.float32)
        if rows > 1:
            x[rows // 2] = 0.0
        t = torch.from_numpy(x).to(DEV)
        o = torch.empty_like(t)
        hip.set_rng_state((31, 4000))
        what = "rows case %d: %s %s %dx%d" % (case, mode, kind, rows, cols)
        if kind == "blocknorm":
            hip.quantize_blocks_(t, o, cols, fmt, rm)
            got = o.cpu().numpy()
            want = orc.cast_blocknorm(x.reshape(-1), mode, cols, seed=31, offset=4000).reshape(rows, cols)
            assert_bits_equal(got, want, what, nan_payload_ok=True)
        else:
            fmax = FLT_MAX[mode.split("_")[0]]
            hip.quantize_rows_(t, o, rows, fmt, rm, fmax, clamp=(kind == "clamp"))
            got = o.cpu().numpy()
            for r in sorted(set([0, rows - 1, rows // 2] + [int(v) for v in rng.integers(0, rows, size=6)])):
                s = orc.scale_from_amax(orc.amax(x[r]), fmax, clamp=(kind == "clamp"))
                want = orc.cast(x[r], mode, scale=s, seed=31, offset=4000 + r)
                assert_bits_equal(got[r], want, what + " row %d" % r, nan_payload_ok=True)
