"""SURVEY 8(f)-4 on the GPU: E4M3_IEEE_{RNE,STOCHASTIC}, FP4_NEAREST and block_norm=True, against the
reference's own outputs (tests/golden/ext2.npz) and the oracle."""
import numpy as np
import pytest
import torch

import orc
from _util import assert_bits_equal, bits, mixed_inputs, npz

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def hip():
    from mpemu.pytquant.hip import _binding as b

    b.lib()
    return b


def _cast(hip, x, mode, scale=1.0, **kw):
    fmt, rm = hip.parse_mode(mode)
    t = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(DEV)
    o = torch.empty_like(t)
    hip.cast_(t, o, fmt, rm, scale=scale, **kw)
    return o.cpu().numpy()


@pytest.mark.parametrize("mode", ["E4M3_IEEE_RNE", "FP4_NEAREST"])
def test_lut_and_raw_vs_reference(hip, mode):
    g = npz("ext2.npz")
    h = np.arange(65536, dtype=np.uint16).view(np.float16).astype(np.float32)
    assert_bits_equal(_cast(hip, h, mode), g["lut/" + mode], mode + " lut", nan_payload_ok=True)
    x = g["input_bits"].view(np.float32)
    for i, sc in enumerate(g["raw_scales"]):
        assert_bits_equal(_cast(hip, x, mode, scale=float(sc)), g["raw/%s/%d" % (mode, i)],
                          "%s scale %r" % (mode, sc), nan_payload_ok=True)


@pytest.mark.parametrize("mode", ["E4M3_IEEE_RNE", "FP4_NEAREST", "E4M3_IEEE_STOCHASTIC"])
@pytest.mark.parametrize("n", [1, 63, 1025, (1 << 20) + 777])
def test_sizes_vs_oracle(hip, mode, n):
    x = mixed_inputs(n, seed=n + 9)
    for s in (1.0, 3.7):
        assert_bits_equal(_cast(hip, x, mode, scale=s, seed=8, offset=3), orc.cast(x, mode, scale=s, seed=8, offset=3),
                          "%s n=%d" % (mode, n), nan_payload_ok=True)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("mode", ["E4M3_IEEE_RNE", "FP4_NEAREST"])
def test_16bit_storage(hip, dt, mode):
    fmt, rm = hip.parse_mode(mode)
    pat = np.arange(65536, dtype=np.uint16)
    t = torch.from_numpy(pat.view(np.int16)).to(DEV).view(dt)
    o = torch.empty_like(t)
    hip.cast_(t, o, fmt, rm, scale=2.0)
    got = o.view(torch.int16).cpu().numpy().view(np.uint16)
    want = orc.cast16(pat, dt == torch.bfloat16, mode, scale=2.0)
    emask, mmask = (0x7F80, 0x007F) if dt == torch.bfloat16 else (0x7C00, 0x03FF)
    nan = lambda a: ((a & emask) == emask) & ((a & mmask) != 0)  # noqa: E731
    bad = np.nonzero((got != want) & ~(nan(got) & nan(want)))[0]
    assert bad.size == 0, (mode, dt, bad[:5], got[bad[:5]], want[bad[:5]])


def test_e4m3_ieee_stochastic_statistics(hip):
    g = npz("ext2.npz")
    N = int(g["sr/N"][0])
    for xb, lo, hi, n_hi in list(zip(g["sr/x_bits"], g["sr/lo_bits"], g["sr/hi_bits"], g["sr/n_hi"]))[::4]:
        x = np.full(N, np.array([xb], dtype=np.uint32).view(np.float32)[0], dtype=np.float32)
        out = bits(_cast(hip, x, "E4M3_IEEE_STOCHASTIC", seed=77, offset=int(xb) & 0xFFF))
        assert np.all((out == lo) | (out == hi))
        p = ((int(xb) >> 13) & 0x7F) / 128.0
        sigma = max(N * p * (1 - p), 1.0) ** 0.5
        assert abs(int((out == hi).sum()) - int(n_hi)) < 7 * sigma
    sub = g["sr/sub_in"].view(np.float32)
    assert_bits_equal(_cast(hip, sub, "E4M3_IEEE_STOCHASTIC", seed=3), g["sr/sub_out"], "sub-range RNE under SR")


def _bn(hip, x, mode, bs, inplace=False):
    t = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(DEV)
    out = hip.forward(t, mode, t.numel(), inplace, 1.0, True, bs)[0]
    assert (out is t) == inplace
    return out.cpu().numpy()


def test_blocknorm_vs_reference(hip):
    g = npz("ext2.npz")
    x = g["input_bits"].view(np.float32)
    for case in g["bn_cases"]:
        mode, bs = str(case).split("/")
        assert_bits_equal(_bn(hip, x, mode, int(bs)), g["bn/" + str(case)], "bn " + str(case), nan_payload_ok=True)
    x24 = x[4096:4096 + 6144]
    for mode in ("E5M2_RNE", "E4M3_RNE", "E3M4_RNE", "FP4_NEAREST"):
        assert_bits_equal(_bn(hip, x24, mode, 24, inplace=True), g["bn24/" + mode], "bn24 " + mode, nan_payload_ok=True)
    assert_bits_equal(_bn(hip, x, "E4M3_RNE", 100), g["bn_demoted/E4M3_RNE/100"], "demoted", nan_payload_ok=True)
    # formats whose reference kernel has no block_norm branch ignore the flag
    assert_bits_equal(_bn(hip, x, "BFLOAT16_RNE", 64), orc.cast(x, "BFLOAT16_RNE"), "bf16 ignores block_norm")
    assert_bits_equal(_bn(hip, x, "E5M2_DAZ_RNE", 64), orc.cast(x, "E5M2_DAZ_RNE"), "daz ignores block_norm",
                      nan_payload_ok=True)


@pytest.mark.parametrize("mode", ["E5M2_RNE", "E4M3_RNE", "E3M4_STOCHASTIC", "E4M3_IEEE_RNE", "FP4_NEAREST",
                                  "E5M2_STOCHASTIC"])
@pytest.mark.parametrize("bs", [1, 2, 16, 17, 100, 1024, 1025, 5000, 40000])
def test_blocknorm_team_shapes_vs_oracle(hip, mode, bs):
    """Block sizes that select the lane / wave / block team, with and without 16-byte rows."""
    nb = 37 if bs < 5000 else 3
    x = mixed_inputs(bs * nb, seed=bs)
    hip.manual_seed(21)
    t = torch.from_numpy(x).to(DEV)
    got = hip.forward(t, mode, t.numel(), False, 1.0, True, bs)[0].cpu().numpy()
    assert_bits_equal(got, orc.cast_blocknorm(x, mode, bs, seed=21, offset=0), "%s bs=%d" % (mode, bs),
                      nan_payload_ok=True)
    if "STOCHASTIC" in mode:
        assert hip.get_rng_state() == (21, nb)  # one call offset per block


@pytest.mark.parametrize("mode", ["E4M3_RNE", "E5M2_STOCHASTIC", "FP4_NEAREST"])
@pytest.mark.parametrize("bs", [1, 2, 4, 8, 12, 32, 64, 128, 256, 260, 576, 2048, 4096, 4100, 8192, 8196])
def test_blocknorm_register_kernel_shapes(hip, mode, bs):
    """16-byte-aligned blocks up to 8192 fp32 elements are held in registers by a lane group of 1..256
    lanes (one read of the data); every group size, partial last slots and the hand-over to the two-pass
    kernel at 8196."""
    nb = 1000 if bs < 4 else 37 if bs <= 2048 else 5   # bs < 4: sub-vector blocks, whole vectors only
    x = mixed_inputs(bs * nb, seed=1000 + bs)
    hip.manual_seed(9)
    t = torch.from_numpy(x).to(DEV)
    got = hip.forward(t, mode, t.numel(), False, 1.0, True, bs)[0].cpu().numpy()
    assert_bits_equal(got, orc.cast_blocknorm(x, mode, bs, seed=9, offset=0), "%s bs=%d" % (mode, bs),
                      nan_payload_ok=True)
    if "STOCHASTIC" in mode:
        assert hip.get_rng_state() == (9, nb)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("bs", [1, 2, 4, 8, 64, 520, 4096, 16384, 16392])
def test_blocknorm_register_kernel_16bit_storage(hip, dt, bs):
    nb = 11 if bs >= 8 else 400
    t = (torch.randn(bs * nb, device=DEV, generator=torch.Generator(device=DEV).manual_seed(bs)) * 5).to(dt)
    got = hip.forward(t, "E4M3_RNE", t.numel(), False, 1.0, True, bs)[0]
    want = torch.from_numpy(orc.cast_blocknorm(t.float().cpu().numpy(), "E4M3_RNE", bs)).to(dt)
    assert torch.equal(got.cpu().view(torch.int16), want.view(torch.int16)), (dt, bs)


def test_emb_norm_call_shape(hip):
    """hybrid_emu.py:119 calls FPEmuOp.apply(param, mode, True, emb_norm, cols): positional slots make that
    scale=True, blocknorm=cols (truthy), blocksize=1 -- block norm with one element per block."""
    from mpemu.pytquant import fpemu_cuda

    x = mixed_inputs(300 * 64, seed=4).reshape(300, 64)
    x[np.isnan(x)] = 0.5
    t = torch.from_numpy(x).to(DEV)
    hip.manual_seed(5)
    y = fpemu_cuda.FPEmuOp.apply(t, "E4M3_STOCHASTIC", True, True, t.size()[1])
    assert y.data_ptr() == t.data_ptr()
    want = orc.cast_blocknorm(x.ravel(), "E4M3_STOCHASTIC", 1, seed=5, offset=0).reshape(x.shape)
    assert_bits_equal(y.cpu().numpy(), want, "emb_norm", nan_payload_ok=True)


def test_blocknorm_large_block1(hip):
    """block_size = 1 over an embedding-table-sized tensor: the lane-team kernel, not one wave per element."""
    n = 30522 * 768
    g = torch.Generator(device=DEV).manual_seed(3)
    t = torch.randn(n, device=DEV, generator=g)
    out = hip.forward(t, "E5M2_RNE", n, False, 1.0, True, 1)[0]
    idx = torch.arange(0, n, 997, device=DEV)
    want = orc.cast_blocknorm(t[idx].cpu().numpy(), "E5M2_RNE", 1)
    assert_bits_equal(out[idx].cpu().numpy(), want, "block 1 samples")
