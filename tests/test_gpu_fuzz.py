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


def test_random_float64_cases_vs_oracle(hip):
    """float64 storage (round 3): random modes, lengths, misaligned views, scales, shard positions, in place."""
    rng = np.random.default_rng(64064)
    for case in range(120):
        mode = ALL_MODES[int(rng.integers(len(ALL_MODES)))]
        fmt, rm = hip.parse_mode(mode)
        n = int(rng.choice([1, 2, 3, 127, 128, 129, 1000, 4097, 65536, 100001]))
        shift = int(rng.integers(0, 3))
        scale = 1.0 if mode.split("_")[0] in ("BFLOAT16", "FLOAT16") else float(rng.choice([1.0, 0.37, 3.0, 116.894]))
        elem_base = int(rng.integers(0, 1 << 20)) * 1024
        seed, offset = int(rng.integers(1 << 40)), int(rng.integers(1 << 20))
        inplace = bool(rng.integers(2))
        x32 = mixed_inputs(n + shift, seed=1000 + case)
        with np.errstate(over="ignore", invalid="ignore"):
            x = x32.astype(np.float64) * (1.0 + rng.standard_normal(n + shift) * 2.0 ** -30)   # bits below fp32's
        t = torch.from_numpy(x).to(DEV)[shift:]
        o = t if inplace else torch.empty(n + shift, dtype=torch.float64, device=DEV)[shift:]
        hip.cast_(t, o, fmt, rm, scale=scale, seed=seed, offset=offset, elem_base=elem_base)
        s32 = np.float32(scale)
        with np.errstate(over="ignore", invalid="ignore"):
            p = (x[shift:] * np.float64(s32)).astype(np.float32)
        q = orc.cast(p, mode, scale=1.0, seed=seed, offset=offset, elem_base=elem_base)
        want = (q * np.float32(1.0 / np.float64(s32))).astype(np.float64)
        got = o.cpu().numpy()
        bad = (got.view(np.uint64) != want.view(np.uint64)) & ~(np.isnan(got) & np.isnan(want))
        assert not bad.any(), "f64 case %d: %s n=%d shift=%d scale=%r inplace=%s: %d differ" % (
            case, mode, n, shift, scale, inplace, int(bad.sum()))


def test_random_16bit_rows_vs_oracle(hip):
    """Per-row quantisation of fp16 / bf16 storage, stochastic modes included (one Philox draw per 8-element vector since
    round 3), over random shapes: the oracle's 16-bit cast of each row with the row's scale."""
    from _util import FLT_MAX

    rng = np.random.default_rng(1616)
    modes = ["E4M3_RNE", "E5M2_STOCHASTIC", "E4M3_STOCHASTIC", "E3M4_STOCHASTIC", "E5M2_RNE"]
    for case in range(80):
        mode = modes[int(rng.integers(len(modes)))]
        fmt, rm = hip.parse_mode(mode)
        cols = int(rng.choice([8, 16, 24, 64, 200, 512, 1000, 1024, 4096, 8200, 16384, 32768, 65536, 70000]))
        rows = int(rng.integers(1, max(2, min(300, (1 << 20) // cols))))
        bf = bool(rng.integers(2))
        dt = torch.bfloat16 if bf else torch.float16
        x = (rng.standard_normal((rows, cols)) * np.exp(rng.standard_normal((rows, 1)) * 2)).astype(np.float32)
        t = torch.from_numpy(x).to(DEV).to(dt)
        o = torch.empty_like(t)
        fmax = FLT_MAX[mode.split("_")[0]]
        hip.set_rng_state((77, 900))
        hip.quantize_rows_(t, o, rows, fmt, rm, fmax, clamp=True)
        bits_in = t.view(torch.int16).cpu().numpy().view(np.uint16)
        got = o.view(torch.int16).cpu().numpy().view(np.uint16)
        up = t.float().cpu().numpy()
        for r in sorted(set([0, rows - 1] + [int(v) for v in rng.integers(0, rows, size=4)])):
            s = orc.scale_from_amax(orc.amax(up[r]), fmax, clamp=True)
            want = orc.cast16(bits_in[r], bf, mode, scale=s, seed=77, offset=900 + r)
            assert np.array_equal(got[r], want), "16-bit rows case %d: %s %s %dx%d row %d" % (
                case, mode, "bf16" if bf else "f16", rows, cols, r)


def test_random_gemm_shapes_with_epilogue_maximum(hip):
    """c-model GEMM over random small shapes, transposes, alpha and accumulate, through the _amax entry point: the
    matrix against the oracle's chains, the word against the matrix."""
    rng = np.random.default_rng(909)
    for case in range(60):
        M, N, K = (int(rng.integers(1, 400)), int(rng.integers(1, 400)), int(rng.choice([1, 3, 16, 32, 64, 96, 130, 512])))
        if case % 6 == 0:
            M, N, K = 4 * int(rng.integers(260, 300)), 4 * int(rng.integers(260, 300)), 32   # the interior + strips split
        at, bt = bool(rng.integers(2)), bool(rng.integers(2))
        variant, fused = ((1, True), (0, False))[int(rng.integers(2))]
        alpha = float(rng.choice([1.0, 0.5, -2.0]))
        A = rng.standard_normal((K, M) if at else (M, K)).astype(np.float32)
        B = rng.standard_normal((N, K) if bt else (K, N)).astype(np.float32)
        C0 = rng.standard_normal((M, N)).astype(np.float32)
        C = torch.from_numpy(C0).to(DEV)
        word = torch.zeros(1, dtype=torch.int32, device=DEV)
        hip.gemm_(C, torch.from_numpy(A).to(DEV), torch.from_numpy(B).to(DEV), alpha, at, bt, variant, amax=word)
        prod = orc.gemm(A, B, 1.0, at, bt, fused=fused)
        want = C0 + (np.float32(alpha) * prod + np.float32(0.0) * C0)
        what = "gemm case %d: %dx%dx%d at=%d bt=%d variant=%d alpha=%g" % (case, M, N, K, at, bt, variant, alpha)
        assert_bits_equal(C.cpu().numpy(), want, what)
        assert int(word.item()) == int(np.abs(want).max().view(np.uint32)), what


def test_random_conv_shapes_vs_oracle(hip):
    """c-model convolution over random geometries -- channel counts on both sides of the 64-channel blocks and of the
    few-channel backward kernel, strides 1-4 (backward-data by phase of the stride, widths that do and do not take
    the 16-byte interleave), rectangular images and filters, padding 0-2 -- forward, backward-data and weight update
    against the oracle's serial chains, both rounding variants."""
    rng = np.random.default_rng(4242)
    done = 0
    while done < 40:
        N = int(rng.integers(1, 4))
        C = int(rng.choice([1, 3, 4, 5, 8, 24, 64, 128, 192]))   # <= 64 or a multiple of 64 (the reference's packing)
        K = int(rng.choice([1, 4, 8, 20, 64, 128]))
        R, S = int(rng.integers(1, 6)), int(rng.integers(1, 6))
        st = int(rng.choice([1, 1, 2, 2, 3, 4]))
        pd = int(rng.integers(0, 3))
        H, W = int(rng.integers(4, 23)), int(rng.choice([4, 7, 8, 10, 12, 13, 16, 20, 21]))
        if H + 2 * pd < R or W + 2 * pd < S:
            continue
        OH, OW = (H + 2 * pd - R) // st + 1, (W + 2 * pd - S) // st + 1
        variant, fused = ((1, True), (0, False))[done % 2]
        x = rng.standard_normal((N, C, H, W)).astype(np.float32)
        w = (rng.standard_normal((K, C, R, S)) * 0.3).astype(np.float32)
        b = rng.standard_normal(K).astype(np.float32)
        go = rng.standard_normal((N, K, OH, OW)).astype(np.float32)
        tx, tw, tgo = (torch.from_numpy(a).to(DEV) for a in (x, w, go))
        y = torch.empty(go.shape, device=DEV)
        hip.conv2d_fp_(y, tx, tw, torch.from_numpy(b).to(DEV), st, pd, variant)
        gi = torch.full(x.shape, 7.0, device=DEV)
        hip.conv2d_bp_(gi, tgo, tw, st, pd, variant)
        gw = torch.full(w.shape, 7.0, device=DEV)
        hip.conv2d_wu_(gw, tgo, tx, st, pd, variant)
        what = "conv case %d: N%d C%d %dx%d K%d %dx%d s%d p%d variant %d" % (done, N, C, H, W, K, R, S, st, pd, variant)
        assert_bits_equal(y.cpu().numpy(), orc.conv2d_fp(x, w, b, st, pd, fused), what + " fp")
        assert_bits_equal(gi.cpu().numpy(), orc.conv2d_bp(go, w, (H, W), st, pd, fused), what + " bp")
        assert_bits_equal(gw.cpu().numpy(), orc.conv2d_wu(go, x, (R, S), st, pd, fused), what + " wu")
        done += 1
