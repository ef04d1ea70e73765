"""GPU parity: the gfx950 HIP cast (through the C ABI) vs the CPU oracle and the reference's
golden vectors.  Bit-exact for every deterministic mode (the bar for integer/bit work);
stochastic modes are bit-exact against the oracle driven by the same Philox stream and
distribution-checked against the reference's scalar routine."""
import math

import numpy as np
import pytest
import torch

import orc
from _util import (ALL_MODES, DET_MODES, EXT_MODES, FLT_MAX, SR_MODES, assert_bits_equal, bits, jload,
                   mixed_inputs, npz)

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(scope="module")
def hip():
    from mpemu.pytquant.hip import _binding as b

    b.lib()
    return b


def gpu_cast(hip, x, mode, scale=1.0, inplace=False, seed=0, offset=0, elem_base=0):
    fmt, rm = hip.parse_mode(mode)
    t = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(DEV)
    o = t if inplace else torch.empty_like(t)
    hip.cast_(t, o, fmt, rm, scale=scale, seed=seed, offset=offset, elem_base=elem_base)
    torch.cuda.synchronize()
    return o.cpu().numpy()


@pytest.mark.parametrize("mode", [m for m in DET_MODES if not m.startswith("BFLOAT16")])
def test_lut_all_fp16_patterns_vs_reference(hip, mode):
    g = npz("luts.npz")
    assert_bits_equal(gpu_cast(hip, g["input_bits"].view(np.float32), mode), g[mode], mode)


@pytest.mark.parametrize("mode", [m for m in DET_MODES + EXT_MODES + SR_MODES if m.split("_")[0] in ("E5M2", "E4M3", "E3M4")
                                  and "DAZ" not in m and "IEEE" not in m])
def test_packed_and_scalar_quantisers_agree_on_all_fp16_patterns(hip, mode):
    """The 16-byte path quantises two fp16 patterns per instruction (quant_h2), the per-lane path (tails,
    unaligned buffers) one (quant_h): all 65 536 patterns, with and without a scale, through both -- the same
    bits, and the oracle's (stochastic modes: the same Philox stream, so the noise is the same too)."""
    fmt, rm = hip.parse_mode(mode)
    g = npz("luts.npz")
    x = g["input_bits"].view(np.float32)
    pad = torch.zeros(x.size + 1, dtype=torch.float32, device=DEV)
    pad[1:] = torch.from_numpy(x).to(DEV)
    src_unaligned = pad[1:]
    assert src_unaligned.data_ptr() % 16 != 0
    for s in (1.0, 0.37, 116.894):
        packed = gpu_cast(hip, x, mode, scale=s, seed=9, offset=5)
        dst = torch.empty(x.size + 1, dtype=torch.float32, device=DEV)[1:]
        hip.cast_(src_unaligned, dst, fmt, rm, scale=s, seed=9, offset=5)
        torch.cuda.synchronize()
        want = orc.cast(x, mode, scale=s, seed=9, offset=5)
        assert_bits_equal(packed, want, "%s packed scale=%r" % (mode, s))
        assert_bits_equal(dst.cpu().numpy(), want, "%s per-lane scale=%r" % (mode, s))
        if s == 1.0 and mode in g.files:
            assert_bits_equal(packed, g[mode], mode + " vs reference LUT")


@pytest.mark.parametrize("mode", DET_MODES)
def test_rawbits_vs_reference(hip, mode):
    g = npz("rawbits.npz")
    x = g["input_bits"].view(np.float32)
    assert_bits_equal(gpu_cast(hip, x, mode), g[mode], mode)
    assert_bits_equal(gpu_cast(hip, x, mode, inplace=True), g[mode], mode + " in-place")


@pytest.mark.parametrize("mode", ["E5M2_RNE", "E4M3_RNE", "E3M4_RNE", "E5M2_RTZ"])
def test_scaled_vs_reference(hip, mode):
    """Includes the mul -> fp16 double-rounding contract (SURVEY section 7 "compiler fusion trap")."""
    g = npz("scaled.npz")
    x = g["input_bits"].view(np.float32)
    for i, s in enumerate(g["scales"]):
        assert_bits_equal(gpu_cast(hip, x, mode, scale=float(s)), g["%s/%d" % (mode, i)],
                          "%s scale=%r" % (mode, s))


def test_reference_testpy_vector(hip):
    d = jload("testpy64.json")
    x = np.array([int(v, 16) for v in d["input_bits"]], dtype=np.uint32).view(np.float32)
    for mode, out in d["modes"].items():
        want = np.array([int(v, 16) for v in out], dtype=np.uint32)
        assert_bits_equal(gpu_cast(hip, x, mode), want, mode)


@pytest.mark.parametrize("mode", DET_MODES + EXT_MODES)
@pytest.mark.parametrize("n", [0, 1, 3, 63, 64, 255, 1023, 1024, 1025, 4099, 1 << 20, (1 << 20) + 777])
def test_sizes_and_tails_vs_oracle(hip, mode, n):
    """Empty, ragged and multi-superblock sizes: vector path + tail path in one launch."""
    x = mixed_inputs(max(n, 1), seed=n + 5)[:n]
    if n == 0:
        assert gpu_cast(hip, x, mode).size == 0
        return
    for s in (1.0, 3.7):
        assert_bits_equal(gpu_cast(hip, x, mode, scale=s), orc.cast(x, mode, scale=s), "%s n=%d" % (mode, n))


@pytest.mark.parametrize("mode", ["E4M3_RNE", "E5M2_RNE", "E3M4_RNE", "BFLOAT16_RNE", "E5M2_STOCHASTIC"])
def test_unaligned_views(hip, mode):
    """Buffers that are only 4-byte aligned (row views of odd-width matrices) take the per-lane path."""
    fmt, rm = hip.parse_mode(mode)
    x = mixed_inputs(3 * 1031 + 5, seed=2)
    t = torch.from_numpy(x).to(DEV)
    for start, length in ((1, 1031), (1033, 1031), (3, 2500), (2, 2)):
        src = t[start:start + length]
        assert src.data_ptr() % 16 != 0
        dst = torch.empty(length + 1, dtype=torch.float32, device=DEV)[1:]
        hip.cast_(src, dst, fmt, rm, scale=2.5, seed=4, offset=2)
        torch.cuda.synchronize()
        want = orc.cast(x[start:start + length], mode, scale=2.5, seed=4, offset=2)
        assert_bits_equal(dst.cpu().numpy(), want, "%s [%d:+%d]" % (mode, start, length))


def test_full_size_config1_e4m3(hip):
    """BASELINE config 1: 4096x4096 randn, E4M3_RNE, scale 1 -- whole tensor vs the oracle."""
    torch.manual_seed(0)
    x = torch.randn(4096, 4096).numpy()
    assert_bits_equal(gpu_cast(hip, x, "E4M3_RNE"), orc.cast(x, "E4M3_RNE"), "config1")


def test_full_size_config1_e4m3_per_tensor_max(hip):
    """BASELINE config 1, second half: the same tensor with the per-tensor max scale derived on device."""
    torch.manual_seed(0)
    x = torch.randn(4096, 4096)
    fmt, rm = hip.parse_mode("E4M3_RNE")
    t = x.to(DEV)
    o = torch.empty_like(t)
    hip.quantize_per_tensor_(t, o, fmt, rm, 448.0)
    s = orc.scale_from_amax(orc.amax(x.numpy()), 448.0)
    assert_bits_equal(o.cpu().numpy(), orc.cast(x.numpy(), "E4M3_RNE", scale=s), "config1 per-tensor max")


@pytest.mark.parametrize("mode", ALL_MODES)
def test_16mi_prefix_all_modes_vs_oracle(hip, mode):
    """BASELINE config 2 input mix, 16 Mi elements, every format x rounding mode."""
    x = mixed_inputs(1 << 24, seed=1234)
    got = gpu_cast(hip, x, mode, seed=42, offset=9)
    want = orc.cast(x, mode, seed=42, offset=9)
    assert_bits_equal(got, want, mode)


@pytest.mark.parametrize("mode", ["E4M3_RNE", "E5M2_RNE", "E3M4_RNE", "BFLOAT16_RNE"])
def test_full_1gib_properties(hip, mode):
    """At BASELINE's full size (1 GiB fp32) the oracle is too slow for a whole-tensor compare:
    check size-independent properties instead -- idempotence (cast(cast(x)) == cast(x)),
    in-place == out-of-place, sign symmetry, and an exact compare on strided samples."""
    fmt, rm = hip.parse_mode(mode)
    n = 1 << 28
    g = torch.Generator(device=DEV).manual_seed(1)
    x = torch.randn(n, device=DEV, generator=g) * torch.exp(torch.randn(n, device=DEV, generator=g) * 3)
    y = torch.empty_like(x)
    hip.cast_(x, y, fmt, rm)
    y2 = torch.empty_like(x)
    hip.cast_(y, y2, fmt, rm)
    assert torch.equal(y.view(torch.int32), y2.view(torch.int32)), "idempotence"
    del y2
    neg = torch.empty_like(x)
    hip.cast_(-x, neg, fmt, rm)
    assert torch.equal((-neg).view(torch.int32), y.view(torch.int32)), "sign symmetry"
    del neg
    xi = x.clone()
    hip.cast_(xi, xi, fmt, rm)
    assert torch.equal(xi.view(torch.int32), y.view(torch.int32)), "in-place"
    del xi
    idx = torch.arange(0, n, 4099, device=DEV)
    tail = torch.arange(n - 70000, n, device=DEV)
    for sel in (idx, tail):
        xs = x[sel].cpu().numpy()
        assert_bits_equal(y[sel].cpu().numpy(), orc.cast(xs, mode), mode + " samples")


def test_forward_signature_and_errors(hip):
    """pybind-style forward(input, mode, size, inplace, scale, block_norm, block_size)."""
    x = torch.from_numpy(mixed_inputs(5000, seed=8)).to(DEV)
    out = hip.forward(x, "E4M3_RNE", x.numel(), False, 2.0, False, 1)
    assert isinstance(out, list) and len(out) == 1 and out[0] is not x
    assert_bits_equal(out[0].cpu().numpy(), orc.cast(x.cpu().numpy(), "E4M3_RNE", scale=2.0))
    xi = x.clone()
    out = hip.forward(xi, "E5M2_RNE", xi.numel(), True, 1.0, False, 1)
    assert out[0] is xi
    assert_bits_equal(xi.cpu().numpy(), orc.cast(x.cpu().numpy(), "E5M2_RNE"))
    with pytest.raises(ValueError):
        hip.forward(x, "E4M3_BOGUS", x.numel(), False, 1.0, False, 1)
    out = hip.forward(x, "E4M3_RNE", x.numel(), False, 1.0, True, 8)   # block norm (tests/test_gpu_ext2.py)
    assert_bits_equal(out[0].cpu().numpy(), orc.cast_blocknorm(x.cpu().numpy(), "E4M3_RNE", 8), nan_payload_ok=True)
    # size % block_size != 0 demotes block_norm (cpp:2020-2025) and the plain cast runs
    out = hip.forward(x, "E4M3_RNE", x.numel(), False, 1.0, True, 7)
    assert_bits_equal(out[0].cpu().numpy(), orc.cast(x.cpu().numpy(), "E4M3_RNE"))
    with pytest.raises(ValueError):
        hip.forward(x, "E4M3_RNE", x.numel(), False, float("inf"), False, 1)
    with pytest.raises(RuntimeError):
        hip.forward(x.cpu(), "E4M3_RNE", x.numel(), False, 1.0, False, 1)
    with pytest.raises(TypeError):
        hip.forward(x.to(torch.int32), "E4M3_RNE", x.numel(), False, 1.0, False, 1)
    with pytest.raises(ValueError):   # float64 storage: plain casts only, no block norm
        hip.forward(x.double(), "E4M3_RNE", x.numel(), False, 1.0, True, 8)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("mode", ["E5M2_RNE", "E4M3_RNE", "E3M4_RNE", "BFLOAT16_RNE", "FLOAT16_RNE",
                                  "E5M2_RTZ", "E4M3_STOCHASTIC", "BFLOAT16_STOCHASTIC"])
def test_16bit_storage_vs_oracle(hip, dt, mode):
    """fp16 / bf16 tensors: up-convert, canonical fp32 algorithm, RNE down-convert
    (SURVEY 8(a) "Non-fp32 inputs"; parity otherwise unpinned by the reference)."""
    fmt, rm = hip.parse_mode(mode)
    rng = np.random.default_rng(6)
    pat = np.concatenate([np.arange(65536, dtype=np.uint16),
                          rng.integers(0, 65536, size=70001, dtype=np.uint16)])
    t = torch.from_numpy(pat.view(np.int16)).to(DEV).view(dt)
    for s in (1.0, 3.0):
        o = torch.empty_like(t)
        hip.cast_(t, o, fmt, rm, scale=s, seed=5, offset=1)
        torch.cuda.synchronize()
        got = o.view(torch.int16).cpu().numpy().view(np.uint16)
        want = orc.cast16(pat, dt == torch.bfloat16, mode, scale=s, seed=5, offset=1)
        # NaN in, NaN out: which payload survives an fp16->fp32->fp16 round trip of a SIGNALLING
        # NaN is an ISA/compiler convention (hipcc folds the pair away and keeps it signalling,
        # a convert instruction quiets it); any NaN pattern is accepted there.
        emask, mmask = (0x7F80, 0x007F) if dt == torch.bfloat16 else (0x7C00, 0x03FF)

        def isnan(a):
            return ((a & emask) == emask) & ((a & mmask) != 0)

        bad = np.nonzero((got != want) & ~(isnan(got) & isnan(want)))[0]
        assert bad.size == 0, (mode, dt, s, bad[:5], got[bad[:5]], want[bad[:5]], pat[bad[:5]])


# ------------------------------------------------------------------------------------------
# amax / device-side scale / fused per-tensor quantisation
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [1, 5, 1023, 4096, 100003, 1 << 22])
def test_amax_vs_oracle(hip, n):
    x = mixed_inputs(n, seed=n)
    x[np.isnan(x)] = 1.0
    for arr in (x, x * 1e-30, np.zeros(n, dtype=np.float32)):
        t = torch.from_numpy(arr).to(DEV)
        ws = torch.zeros(1, dtype=torch.int32, device=DEV)
        hip.amax_(t, ws)
        assert int(ws.item()) == int(bits(np.array([orc.amax(arr)], dtype=np.float32))[0])
    if n > 4:
        x[n // 2] = np.nan
        ws = torch.zeros(1, dtype=torch.int32, device=DEV)
        hip.amax_(torch.from_numpy(x).to(DEV), ws)
        assert math.isnan(ws.view(torch.float32).item())  # torch.max propagates NaN (qutils.py:283)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_amax_16bit(hip, dt):
    t = (torch.randn(100003, device=DEV) * 7).to(dt)
    ws = torch.zeros(1, dtype=torch.int32, device=DEV)
    hip.amax_(t, ws)
    assert ws.view(torch.float32).item() == t.float().abs().max().item()
    ws.zero_()
    hip.amax_(t[1:50000], ws)  # 2-byte aligned view
    assert ws.view(torch.float32).item() == t[1:50000].float().abs().max().item()


@pytest.mark.parametrize("fmtname", ["E4M3", "E5M2", "E3M4"])
@pytest.mark.parametrize("mag", [1.0, 1e-3, 300.0, 1e-7, 0.0, 1e30, 1e-42])
def test_fused_per_tensor_scale_vs_oracle(hip, fmtname, mag):
    """quantize_tensor's per-tensor max policy with the scale derived on device (no host sync):
    scale = fp32(fp32(1/vmax) * flt_max), clamp rule, rcp = fp32(1/double(scale))."""
    mode = fmtname + "_RNE"
    fmt, rm = hip.parse_mode(mode)
    rng = np.random.default_rng(12)
    x = (rng.standard_normal(50000) * mag).astype(np.float32)
    t = torch.from_numpy(x).to(DEV)
    o = torch.empty_like(t)
    hip.quantize_per_tensor_(t, o, fmt, rm, FLT_MAX[fmtname])
    s = orc.scale_from_amax(orc.amax(x), FLT_MAX[fmtname])
    assert_bits_equal(o.cpu().numpy(), orc.cast(x, mode, scale=s), "%s mag=%g scale=%r" % (mode, mag, s))


def test_fused_per_tensor_nan_input(hip):
    fmt, rm = hip.parse_mode("E4M3_RNE")
    x = np.array([1.0, np.nan, -2.0, 0.5] * 300, dtype=np.float32)
    t = torch.from_numpy(x).to(DEV)
    o = torch.empty_like(t)
    hip.quantize_per_tensor_(t, o, fmt, rm, 448.0)
    want = orc.cast(x, "E4M3_RNE", scale=orc.scale_from_amax(orc.amax(x), 448.0))
    got = o.cpu().numpy()
    assert np.all(np.isnan(got)) and np.all(np.isnan(want))  # NaN scale -> all-NaN output


def test_device_scale_random_amax_sweep(hip):
    """The device-side division / reciprocal must be the IEEE ones for every amax."""
    fmt, rm = hip.parse_mode("E4M3_RNE")
    rng = np.random.default_rng(99)
    base = rng.standard_normal(2048).astype(np.float32)
    for k in range(200):
        m = np.float32(np.exp(rng.uniform(-60, 60)))
        x = (base * m).astype(np.float32)
        t = torch.from_numpy(x).to(DEV)
        o = torch.empty_like(t)
        hip.quantize_per_tensor_(t, o, fmt, rm, 448.0)
        s = orc.scale_from_amax(orc.amax(x), 448.0)
        assert_bits_equal(o.cpu().numpy(), orc.cast(x, "E4M3_RNE", scale=s), "amax=%r" % orc.amax(x))


# ------------------------------------------------------------------------------------------
# rows kernel: per-channel / fine-grained scaling in one launch
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("rows,cols", [(1, 1), (3, 7), (64, 147), (17, 1024), (768, 768), (5, 70001),
                                       (3072, 768), (2, 262144), (1000, 36), (37, 2304), (9, 8192),
                                       # the 1024-thread register-resident form: 4 / 8 / 16 vectors per lane,
                                       # whole and ragged last slots, LLM-shaped rows, the edge to the two-pass form
                                       (300, 11008), (40, 14336), (33, 16384), (7, 8196), (6, 40000), (5, 65536),
                                       (3, 65540), (3, 32772)])
@pytest.mark.parametrize("mode,clamp", [("E4M3_RNE", True), ("E5M2_RNE", True), ("E3M4_RNE", False),
                                        ("E5M2_STOCHASTIC", True), ("E4M3_RTZ", False)])
def test_rows_kernel_vs_oracle_loop(hip, rows, cols, mode, clamp):
    fmt, rm = hip.parse_mode(mode)
    fmax = FLT_MAX[mode.split("_")[0]]
    rng = np.random.default_rng(rows * 131 + cols)
    x = (rng.standard_normal((rows, cols)) * np.exp(rng.standard_normal((rows, 1)) * 6)).astype(np.float32)
    if rows > 2:
        x[1] = 0.0                      # all-zero row: vmax = 0 -> inf -> clamp (or inf scale unclamped)
        x[2, 0] = 1e-41                 # denormal maximum candidates
    t = torch.from_numpy(x).to(DEV)
    o = torch.empty_like(t)
    hip.set_rng_state((77, 1000))
    hip.quantize_rows_(t, o, rows, fmt, rm, fmax, clamp=clamp)
    got = o.cpu().numpy()
    sel = range(rows) if rows <= 64 else list(range(0, rows, 97)) + [rows - 1]
    for r in sel:
        s = orc.scale_from_amax(orc.amax(x[r]), fmax, clamp=clamp)
        want = orc.cast(x[r], mode, scale=s, seed=77, offset=1000 + r)
        assert_bits_equal(got[r], want, "%s row %d" % (mode, r), nan_payload_ok=True)
    if rm == hip.STOCHASTIC:
        assert hip.get_rng_state() == (77, 1000 + rows)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_rows_kernel_16bit_storage(hip, dt):
    fmt, rm = hip.parse_mode("E4M3_RNE")
    for rows, cols in ((9, 33), (40, 256), (20, 16392), (11, 28672), (5, 65536), (3, 65544), (2, 131072)):
        t = (torch.randn(rows, cols, device=DEV) * 3).to(dt)
        o = torch.empty_like(t)
        hip.quantize_rows_(t, o, rows, fmt, rm, 448.0)
        x = t.float().cpu().numpy()
        for r in range(rows):
            s = orc.scale_from_amax(orc.amax(x[r]), 448.0)
            want = torch.from_numpy(orc.cast(x[r], "E4M3_RNE", scale=s)).to(dt)
            assert torch.equal(o[r].cpu().view(torch.int16), want.view(torch.int16)), (dt, rows, cols, r)


@pytest.mark.parametrize("mode", ["E5M2_RNE", "E4M3_STOCHASTIC", "E3M4_RTZ"])
@pytest.mark.parametrize("rows,cols", [(64, 1), (32, 2), (16, 4), (7, 13), (40, 64), (33, 768), (9, 4096), (5, 11008),
                                       (3, 16384), (3, 40001), (2, 65536), (300, 100)])
def test_cast_rows_scaled_vs_oracle_loop(hip, rows, cols, mode):
    """fpemu_cast_rows_scaled (one scale per row from device memory) = a loop of per-row casts with
    consecutive call offsets (reference qutils.py:289-311 with the scale of :294-301), every kernel form."""
    fmt, rm = hip.parse_mode(mode)
    rng = np.random.default_rng(rows * 977 + cols)
    x = (rng.standard_normal((rows, cols)) * np.exp(rng.standard_normal((rows, 1)) * 3)).astype(np.float32)
    sc = np.exp(rng.standard_normal(rows) * 3).astype(np.float32)
    sc[0] = 1.0
    t = torch.from_numpy(x).to(DEV)
    scales = torch.from_numpy(sc).to(DEV)
    o = torch.empty_like(t)
    hip.set_rng_state((31, 500))
    hip.cast_rows_scaled_(t, o, rows, fmt, rm, scales)
    got = o.cpu().numpy()
    for r in range(rows):
        want = orc.cast(x[r], mode, scale=float(sc[r]), seed=31, offset=500 + r)
        assert_bits_equal(got[r], want, "%s row %d of %dx%d" % (mode, r, rows, cols), nan_payload_ok=True)
    if rm == hip.STOCHASTIC:
        assert hip.get_rng_state() == (31, 500 + rows)
    # in place, and the scales untouched
    hip.set_rng_state((31, 500))
    hip.cast_rows_scaled_(t, t, rows, fmt, rm, scales)
    assert torch.equal(t.view(torch.int32), o.view(torch.int32))
    assert torch.equal(scales.cpu(), torch.from_numpy(sc))


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_cast_rows_scaled_16bit_storage(hip, dt):
    fmt, rm = hip.parse_mode("E4M3_RNE")
    for rows, cols in ((16, 4), (9, 33), (40, 256), (6, 16392), (3, 65536)):
        t = (torch.randn(rows, cols, device=DEV) * 3).to(dt)
        sc = (torch.rand(rows, device=DEV) * 20 + 0.25).float()
        o = torch.empty_like(t)
        hip.cast_rows_scaled_(t, o, rows, fmt, rm, sc)
        x = t.float().cpu().numpy()
        for r in range(rows):
            want = torch.from_numpy(orc.cast(x[r], "E4M3_RNE", scale=float(sc[r].item()))).to(dt)
            assert torch.equal(o[r].cpu().view(torch.int16), want.view(torch.int16)), (dt, rows, cols, r)


@pytest.mark.parametrize("mode", ["E4M3_RNE", "E5M2_STOCHASTIC", "BFLOAT16_RNE"])
@pytest.mark.parametrize("n", [1, 1000, 1 << 20, (1 << 20) + 3])
def test_cast_device_scale_equals_host_scale(hip, mode, n):
    """fpemu_cast_device_scale(*scale) = fpemu_cast(scale) bit for bit (incl. the noise stream and elem_base)."""
    fmt, rm = hip.parse_mode(mode)
    t = torch.randn(n, device=DEV) * 5
    for sc in (1.0, 0.37, 4096.0, 3.0e-3):
        a, b = torch.empty_like(t), torch.empty_like(t)
        hip.cast_(t, a, fmt, rm, scale=sc, seed=9, offset=77, elem_base=4096)
        hip.cast_device_scale_(t, b, fmt, rm, torch.tensor([sc], dtype=torch.float32, device=DEV), seed=9, offset=77,
                               elem_base=4096)
        assert torch.equal(a.view(torch.int32), b.view(torch.int32)), (mode, n, sc)
    assert_bits_equal(b.cpu().numpy()[:4096], orc.cast(t.cpu().numpy(), mode, scale=3.0e-3, seed=9, offset=77,
                                                        elem_base=4096)[:4096], mode)


def test_cast_rows_scaled_argument_errors(hip):
    t = torch.randn(4, 8, device=DEV)
    sc = torch.ones(4, device=DEV)
    with pytest.raises(ValueError):  # BF16 takes no scale
        hip.cast_rows_scaled_(t, torch.empty_like(t), 4, *hip.parse_mode("BFLOAT16_RNE"), sc)
    with pytest.raises(ValueError):  # too few scales
        hip.cast_rows_scaled_(t, torch.empty_like(t), 4, *hip.parse_mode("E4M3_RNE"), sc[:3])
    with pytest.raises(ValueError):  # wrong dtype
        hip.cast_rows_scaled_(t, torch.empty_like(t), 4, *hip.parse_mode("E4M3_RNE"), sc.double())
    with pytest.raises(ValueError):  # host scales
        hip.cast_device_scale_(t, torch.empty_like(t), *hip.parse_mode("E4M3_RNE"), torch.ones(1))


def test_rows_kernel_rejects_scale_free_formats(hip):
    t = torch.randn(4, 8, device=DEV)
    fmt, rm = hip.parse_mode("BFLOAT16_RNE")
    with pytest.raises(ValueError):
        hip.quantize_rows_(t, torch.empty_like(t), 4, fmt, rm, 1.0)
    with pytest.raises(ValueError):
        hip.quantize_rows_(t, torch.empty_like(t), 5, *hip.parse_mode("E4M3_RNE"), 448.0)


@pytest.mark.parametrize("mode", ["E4M3_RNE", "E5M2_STOCHASTIC", "E3M4_STOCHASTIC"])
@pytest.mark.parametrize("n", [1, 1000, 32767, 32768, 32769, 70000])
def test_fused_per_tensor_small_and_large_paths_agree(hip, mode, n):
    """<= 32768 elements take the one-workgroup fused path, larger ones amax + cast launches: both
    must give the oracle's bits (incl. the stochastic noise) and publish the amax word."""
    fmt, rm = hip.parse_mode(mode)
    fmax = FLT_MAX[mode.split("_")[0]]
    x = (np.random.default_rng(n).standard_normal(n) * 3).astype(np.float32)
    t = torch.from_numpy(x).to(DEV)
    o = torch.empty_like(t)
    ws = torch.full((1,), 12345, dtype=torch.int32, device=DEV)
    hip.set_rng_state((5, 40))
    hip.quantize_per_tensor_(t, o, fmt, rm, fmax, amax_ws=ws)
    s = orc.scale_from_amax(orc.amax(x), fmax)
    assert_bits_equal(o.cpu().numpy(), orc.cast(x, mode, scale=s, seed=5, offset=40), "%s n=%d" % (mode, n))
    assert ws.view(torch.float32).item() == float(orc.amax(x))


def test_over_2g_elements_64bit_indexing(hip):
    """BASELINE config 5's shard: 8 GiB fp32 = 2^31 elements per GPU, beyond the reference's `int size`.
    In-place so that one 8.5 GiB buffer suffices; exact compare on windows at the start, across the 2^31
    and 2^32-byte boundaries and at the ragged end, deterministic and stochastic."""
    n = (1 << 31) + 4096 + 5
    g = torch.Generator(device=DEV).manual_seed(9)
    x = torch.empty(n, device=DEV)
    step = 1 << 28
    for i in range(0, n, step):
        m = min(step, n - i)
        x[i:i + m] = torch.randn(m, device=DEV, generator=g) * torch.exp(torch.randn(m, device=DEV, generator=g) * 2)
    wins = [(0, 5000), ((1 << 30) - 2500, 5000), ((1 << 31) - 2500, 5000), (n - 6000, 6000)]
    keep = [x[a:a + b].cpu().numpy().copy() for a, b in wins]
    for mode in ("E4M3_RNE", "E5M2_STOCHASTIC"):
        fmt, rm = hip.parse_mode(mode)
        y = x.clone() if mode == "E4M3_RNE" else x
        hip.cast_(y, y, fmt, rm, scale=2.0, seed=17, offset=4)
        torch.cuda.synchronize()
        for (a, b), src in zip(wins, keep):
            want = orc.cast(src, mode, scale=2.0, seed=17, offset=4, elem_base=a)
            assert_bits_equal(y[a:a + b].cpu().numpy(), want, "%s window @%d" % (mode, a), nan_payload_ok=True)
        ws = torch.zeros(1, dtype=torch.int32, device=DEV)
        if mode == "E4M3_RNE":
            hip.amax_(x, ws)
            assert ws.view(torch.float32).item() == max(float(x[i:i + step].abs().max()) for i in range(0, n, step))
        del y


@pytest.mark.parametrize("n", [32769, 100003, 786432, (1 << 22), (1 << 22) + 1, 5000000])
@pytest.mark.parametrize("mode", ["E4M3_RNE", "E5M2_STOCHASTIC"])
def test_per_tensor_workspace_path_equals_single_word_path(hip, n, mode):
    """fpemu_quantize_per_tensor_ws (two launches through per-block maxima for mid-size tensors) gives the
    bits of the single-word path and of the oracle, publishes the amax in ws[0], needs no initialised
    scratch, and works on a 4-byte-aligned view."""
    fmt, rm = hip.parse_mode(mode)
    fmax = FLT_MAX[mode.split("_")[0]]
    x = mixed_inputs(n + 1, seed=n % 1000)
    x[np.isnan(x)] = 2.5
    for start in (0, 1):
        hx = x[start:start + n]
        t = torch.from_numpy(x).to(DEV)[start:start + n]
        want = orc.cast(hx, mode, scale=orc.scale_from_amax(orc.amax(hx), fmax), seed=5, offset=0)
        ws = torch.full((513,), 0x7F7FFFFF, dtype=torch.int32, device=DEV)     # garbage scratch
        o1 = torch.empty(n, device=DEV)
        hip.manual_seed(5)
        hip.quantize_per_tensor_(t, o1, fmt, rm, fmax, amax_ws=ws)
        assert_bits_equal(o1.cpu().numpy(), want, "%s n=%d start=%d workspace path" % (mode, n, start))
        assert ws[:1].view(torch.float32).item() == float(orc.amax(hx))
        w1 = torch.zeros(1, dtype=torch.int32, device=DEV)
        o2 = torch.empty(n, device=DEV)
        hip.manual_seed(5)
        hip.quantize_per_tensor_(t, o2, fmt, rm, fmax, amax_ws=w1)
        assert torch.equal(o1.view(torch.int32), o2.view(torch.int32))
        hip.manual_seed(5)
        o3 = torch.empty(n, device=DEV)
        hip.quantize_per_tensor_(t, o3, fmt, rm, fmax)                          # cached per-stream scratch
        assert torch.equal(o1.view(torch.int32), o3.view(torch.int32))


# ------------------------------------------------------------------------------------------
# multi-tensor cast: one launch for a list of buffers == the loop of single casts
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mode", ["FLOAT16_STOCHASTIC", "E4M3_RNE", "E5M2_STOCHASTIC", "BFLOAT16_RNE"])
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_cast_multi_equals_loop_of_casts(hip, mode, dt):
    fmt, rm = hip.parse_mode(mode)
    sizes = [1, 0, 7, 64, 1000, 4096, 65536 + 3, 3 * 7 * 7 * 64, 1 << 20, 5, 0, 2048] + [33 + 17 * i for i in range(60)]
    g = torch.Generator(device=DEV).manual_seed(5)
    base = [(torch.randn(n + 3, device=DEV, generator=g) * 4).to(dt) for n in sizes]
    # every third tensor is a view that is not 16-byte aligned (element path inside the same launch)
    srcs = [b[1:1 + n] if i % 3 == 2 else b[:n] for i, (b, n) in enumerate(zip(base, sizes))]
    want = []
    hip.set_rng_state((11, 500))
    for t in srcs:
        want.append(hip.cast_(t, torch.empty_like(t), fmt, rm, scale=1.0))
    state_after_loop = hip.get_rng_state()
    hip.set_rng_state((11, 500))
    outs = [torch.empty_like(t) for t in srcs]
    hip.cast_multi_(srcs, fmt, rm, dsts=outs)
    assert hip.get_rng_state() == state_after_loop
    for i, (a, b) in enumerate(zip(outs, want)):
        assert torch.equal(a.view(torch.int16 if dt != torch.float32 else torch.int32),
                           b.view(torch.int16 if dt != torch.float32 else torch.int32)), (mode, dt, i, sizes[i])
    # in place, and against the oracle for one deterministic case
    clones = [t.clone() for t in srcs]
    hip.set_rng_state((11, 500))
    hip.cast_multi_(clones, fmt, rm)
    for i, (a, b) in enumerate(zip(clones, want)):
        assert torch.equal(a.view(torch.uint8), b.view(torch.uint8)), (mode, dt, i)
    if mode == "E4M3_RNE" and dt == torch.float32:
        assert_bits_equal(outs[8].cpu().numpy(), orc.cast(srcs[8].cpu().numpy(), mode), "multi vs oracle")


def test_cast_multi_rejects_mixed_lists(hip):
    a = torch.randn(8, device=DEV)
    with pytest.raises(ValueError):
        hip.cast_multi_([a, a.half()], *hip.parse_mode("E4M3_RNE"))
    with pytest.raises(ValueError):
        hip.cast_multi_([a], *hip.parse_mode("E4M3_RNE"), dsts=[torch.empty(9, device=DEV)])
    with pytest.raises(RuntimeError):
        hip.cast_multi_([a, torch.randn(8)], *hip.parse_mode("E4M3_RNE"))


@pytest.mark.parametrize("mode", ["E4M3_RNE", "E5M2_STOCHASTIC", "E3M4_RNE"])
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_quantize_per_tensor_multi_equals_the_loop(hip, mode, dt):
    """fpemu_quantize_per_tensor_multi: tensor i == fpemu_quantize_per_tensor at call offset + i (bits, amax words,
    offsets consumed), over ragged sizes, an empty tensor, unaligned views, more than 32 tensors (two launch
    groups), tensors on either side of the one-workgroup size, and a workspace at its minimum size."""
    fmt, rm = hip.parse_mode(mode)
    fmax = FLT_MAX[mode.split("_")[0]]
    g = torch.Generator(device=DEV).manual_seed(3)
    sizes = [1, 7, 0, 1000, 4097, 32768, 32769, 70001, 1 << 20, (1 << 21) + 5] + [257 * (i + 1) for i in range(30)]
    srcs = []
    for i, n in enumerate(sizes):
        base = (torch.randn(n + 3, device=DEV, generator=g) * (0.01 + 3 * i)).to(dt)
        srcs.append(base[1:1 + n] if i % 4 == 1 else base[:n])       # every fourth: not 16-byte aligned
    srcs[5][17] = 1e4
    want, words = [], []
    hip.set_rng_state((21, 300))
    for t in srcs:
        o = torch.empty_like(t)
        ws1 = torch.zeros(1, dtype=torch.int32, device=DEV)
        hip.quantize_per_tensor_(t, o, fmt, rm, fmax, amax_ws=ws1)
        want.append(o)
        words.append(int(ws1.item()))
    used = hip.get_rng_state()
    for ws_words in (None, len(srcs) + 32):
        hip.set_rng_state((21, 300))
        dsts = [torch.empty_like(t) for t in srcs]
        ws = None if ws_words is None else torch.empty(ws_words, dtype=torch.int32, device=DEV)
        _, ws = hip.quantize_per_tensor_multi_(srcs, fmt, rm, fmax, dsts=dsts, ws=ws)
        assert hip.get_rng_state() == used
        for i, (a, b) in enumerate(zip(dsts, want)):
            v = torch.int32 if dt == torch.float32 else torch.int16
            assert torch.equal(a.view(v), b.view(v)), (mode, dt, i, sizes[i], ws_words)
            if sizes[i]:
                assert int(ws[i].item()) == words[i], (i, sizes[i])
    # in place
    hip.set_rng_state((21, 300))
    clones = [t.clone() for t in srcs]
    hip.quantize_per_tensor_multi_(clones, fmt, rm, fmax)
    for a, b in zip(clones, want):
        assert torch.equal(a.view(torch.int16), b.view(torch.int16))


def test_quantize_per_tensor_multi_argument_errors(hip):
    t = [torch.randn(100, device=DEV) for _ in range(3)]
    with pytest.raises(ValueError):  # BF16 takes no scale
        hip.quantize_per_tensor_multi_(t, *hip.parse_mode("BFLOAT16_RNE"), 1.0)
    with pytest.raises(ValueError):  # workspace too small
        hip.quantize_per_tensor_multi_(t, *hip.parse_mode("E4M3_RNE"), 448.0, ws=torch.empty(5, dtype=torch.int32, device=DEV))
    with pytest.raises(ValueError):  # mixed dtypes
        hip.quantize_per_tensor_multi_([t[0], t[1].half()], *hip.parse_mode("E4M3_RNE"), 448.0)


def _assert_f64_equal(got, want, what):
    """Bit-for-bit equality of two float64 arrays; positions where both sides are NaN count as equal."""
    g = np.ascontiguousarray(got, np.float64).ravel()
    w = np.ascontiguousarray(want, np.float64).ravel()
    bad = np.nonzero((g != w) & ~(np.isnan(g.view(np.float64)) & np.isnan(w.view(np.float64))))[0]
    assert bad.size == 0, "%s: %d / %d differ; first at %s got %s want %s" % (
        what, bad.size, g.size, bad[:6].tolist(), [hex(v) for v in g[bad[:6]]], [hex(v) for v in w[bad[:6]]])


def _oracle_f64(x64, mode, scale=1.0, seed=0, offset=0, elem_base=0):
    """float64 storage as the reference's GPU kernels treat it (hip/fpemu_kernels.hip:429-431, 472): ONE rounding of
    the double product x * scale to fp32, the fp32 algorithm on it, the fp32 product with the reciprocal, widened."""
    s32 = np.float32(scale)
    with np.errstate(over="ignore", invalid="ignore"):
        p = (np.asarray(x64, np.float64) * np.float64(s32)).astype(np.float32)
    q = orc.cast(p, mode, scale=1.0, seed=seed, offset=offset, elem_base=elem_base)
    rcp = np.float32(1.0 / np.float64(s32))
    return (q * rcp).astype(np.float64)


@pytest.mark.parametrize("mode", ["E5M2_RNE", "E5M2_RTZ", "E5M2_STOCHASTIC", "E4M3_RNE", "E4M3_STOCHASTIC", "E3M4_RNE",
                                  "BFLOAT16_RNE", "BFLOAT16_STOCHASTIC", "FLOAT16_RNE", "E5M2_DAZ_RNE", "E4M3_IEEE_RNE",
                                  "FP4_NEAREST"])
def test_float64_storage_vs_oracle(hip, mode):
    """double tensors, which the reference's GPU dispatcher accepts (hip/fpemu_kernels.hip:1070; its CPU path is
    fp32-only, so parity is otherwise unpinned): whole vectors, an odd tail, an unaligned view, a scale, in place, and
    values that need the double -> float rounding (beyond fp32's precision and range)."""
    rng = np.random.default_rng(64)
    n = 100003
    x = rng.standard_normal(n) * np.exp(rng.standard_normal(n) * 4)
    x[:8] = [0.0, -0.0, np.inf, -np.inf, np.nan, 1e300, -1e-300, 1.0 + 2.0 ** -30]
    x[8:16] = [448.0, 57344.0, 464.0 - 2.0 ** -40, 1.0625 + 2.0 ** -45, 2.0 ** -150, 3.0e38, 3.5e38, -2.0 ** -9]
    fmt, rm = hip.parse_mode(mode)
    t = torch.from_numpy(x).to(DEV)
    for scale in (1.0, 3.7):
        if scale != 1.0 and mode.split("_")[0] in ("BFLOAT16", "FLOAT16"):
            continue
        y = torch.empty_like(t)
        hip.cast_(t, y, fmt, rm, scale=scale, seed=9, offset=4)
        want = _oracle_f64(x, mode, scale, seed=9, offset=4)
        _assert_f64_equal(y.cpu().numpy(), want, "%s f64 scale %g" % (mode, scale))
    # an unaligned (8- but not 16-byte) view with elem_base, in place
    buf = torch.empty(n + 1, dtype=torch.float64, device=DEV)
    buf[1:].copy_(t)
    view = buf[1:]
    assert view.data_ptr() % 16 == 8
    hip.cast_(view, view, fmt, rm, scale=1.0, seed=9, offset=4, elem_base=1 << 20)
    _assert_f64_equal(view.cpu().numpy(),
                      _oracle_f64(x, mode, 1.0, seed=9, offset=4, elem_base=1 << 20),
                      mode + " f64 unaligned in place")


def test_float64_through_the_operator_and_quantize_tensor(hip):
    """FPEmuOp.apply / forward() / quantize_tensor with double tensors: plain casts run in the float64 kernel; per-tensor
    scaling computes max and scale with torch in double and hands the scale over as a float, as the reference's Python
    does (qutils.py:282-287) -- no device-side scale for this width."""
    from mpemu import qutils
    from mpemu.pytquant import fpemu_cuda

    rng = np.random.default_rng(65)
    x = rng.standard_normal((37, 53)) * 5
    t = torch.from_numpy(x).to(DEV)
    y = fpemu_cuda.FPEmuOp.apply(t, "E4M3_RNE", False, 2.0)
    assert y.dtype == torch.float64
    _assert_f64_equal(y.cpu().numpy(), _oracle_f64(x, "E4M3_RNE", 2.0), "FPEmuOp f64")
    out = hip.forward(t, "E5M2_RNE", t.numel(), False, 1.0, False, 1)[0]
    _assert_f64_equal(out.cpu().numpy(), _oracle_f64(x, "E5M2_RNE"), "forward f64")
    cfg = qutils.TensorQuantConfig("e4m3", "rne", "per-tensor")
    q = qutils.quantize_tensor(t, cfg, None, inplace=False)
    scale = 448.0 / np.abs(x).max()               # double arithmetic, then a C float at the boundary
    scale = 6.55e4 if scale > 3.275e4 else scale
    _assert_f64_equal(q.cpu().numpy(), _oracle_f64(x, "E4M3_RNE", np.float32(scale)),
                      "quantize_tensor f64 per-tensor")
    qc = qutils.quantize_tensor(t, qutils.TensorQuantConfig("e5m2", "rne", "per-channel"), None, inplace=False)
    for r in (0, 36):
        s = 57344.0 / np.abs(x[r]).max()
        s = 6.55e4 if s > 3.275e4 else s
        _assert_f64_equal(qc[r].cpu().numpy(), _oracle_f64(x[r], "E5M2_RNE", np.float32(s)),
                          "quantize_tensor f64 per-channel row %d" % r)
    with pytest.raises(ValueError):
        hip.amax_(t, torch.zeros(1, dtype=torch.int32, device=DEV))
    with pytest.raises(ValueError):
        hip.quantize_per_tensor_(t, torch.empty_like(t), *hip.parse_mode("E4M3_RNE"), 448.0)
