"""Pins the CPU oracle (oracle/fpemu_oracle.c) against fixtures produced by the REFERENCE's own
fpemu_cpp extension and Python package (tests/golden/make_golden.py).  CPU only."""
import math

import numpy as np
import pytest

import orc
from _util import DET_MODES, EXT_MODES, FLT_MAX, FLT_MIN, assert_bits_equal, bits, jload, npz


def test_philox_known_answers():
    # Random123 kat_vectors for philox4x32-10
    assert [hex(v) for v in orc.philox((0, 0), (0, 0, 0, 0))] == \
        ['0x6627e8d5', '0xe169c58d', '0xbc57ac4c', '0x9b00dbd8']
    assert [hex(v) for v in orc.philox((0xffffffff, 0xffffffff), (0xffffffff,) * 4)] == \
        ['0x408f276d', '0x41c83b0e', '0xa20bc7c6', '0x6d5451fd']
    assert [hex(v) for v in orc.philox((0xa4093822, 0x299f31d0),
                                        (0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344))] == \
        ['0xd16cfe09', '0x94fdcceb', '0x5001e420', '0x24126ea1']


def test_f16_conversions_match_numpy():
    rng = np.random.default_rng(3)
    raw = rng.integers(0, 2 ** 32, size=200000, dtype=np.uint64).astype(np.uint32).view(np.float32)
    L = orc.lib()
    with np.errstate(all="ignore"):
        want = raw.astype(np.float16).view(np.uint16)
    got = np.array([L.orc_f32_to_f16_rne(float(v)) for v in raw[:20000]], dtype=np.uint16)
    nan = np.isnan(raw[:20000])
    assert np.array_equal(got[~nan], want[:20000][~nan])
    h = np.arange(65536, dtype=np.uint16)
    back = np.array([L.orc_f16_to_f32(int(v)) for v in h], dtype=np.float32)
    ref = h.view(np.float16).astype(np.float32)
    keep = ~np.isnan(ref)
    assert np.array_equal(bits(back)[keep], bits(ref)[keep])


@pytest.mark.parametrize("mode", [m for m in DET_MODES if not m.startswith("BFLOAT16")])
def test_lut_all_fp16_patterns(mode):
    g = npz("luts.npz")
    x = g["input_bits"].view(np.float32)
    assert_bits_equal(orc.cast(x, mode), g[mode], mode)


def test_lut_distinct_code_counts():
    # the reference's E4M3 / E3M4 are not 8-bit in the subnormal range
    g = npz("luts.npz")
    x = g["input_bits"].view(np.float32)
    for mode, n in (("E5M2_RNE", 254), ("E4M3_RNE", 288), ("E3M4_RNE", 352)):
        assert np.unique(bits(orc.cast(x, mode))).size == n, mode


def test_lut_sha256_matches_the_survey_probe():
    """SURVEY.md section 4 records sha256 prefixes of the three RNE look-up tables (65 536 fp16 bit patterns as
    uint16) taken from the reference's fpemu_cpp when the survey was made: an independent pin of the committed
    fixture and of the oracle."""
    import hashlib

    g = npz("luts.npz")
    x = g["input_bits"].view(np.float32)
    for mode, want in (("E5M2_RNE", "b3a6875396b597c6"), ("E4M3_RNE", "8eed355ead53df18"),
                       ("E3M4_RNE", "e14706811ade985d")):
        for name, out in (("fixture", g[mode].view(np.float32)), ("oracle", orc.cast(x, mode))):
            lut = out.astype(np.float16).view(np.uint16)  # every output is exactly representable in fp16
            assert np.array_equal(lut.view(np.float16).astype(np.float32).view(np.uint32)[~np.isnan(out)],
                                  out.view(np.uint32)[~np.isnan(out)])
            assert hashlib.sha256(lut.tobytes()).hexdigest()[:16] == want, (mode, name)


@pytest.mark.parametrize("mode", DET_MODES)
def test_rawbits_and_lognormal(mode):
    g = npz("rawbits.npz")
    assert_bits_equal(orc.cast(g["input_bits"].view(np.float32), mode), g[mode], mode)


@pytest.mark.parametrize("mode", ["E5M2_RNE", "E4M3_RNE", "E3M4_RNE", "E5M2_RTZ"])
def test_scaled(mode):
    g = npz("scaled.npz")
    x = g["input_bits"].view(np.float32)
    for i, s in enumerate(g["scales"]):
        assert_bits_equal(orc.cast(x, mode, scale=s), g["%s/%d" % (mode, i)], "%s scale=%r" % (mode, s))


def test_reference_testpy_vector():
    d = jload("testpy64.json")
    x = np.array([int(v, 16) for v in d["input_bits"]], dtype=np.uint32).view(np.float32)
    assert x.size == 64
    for mode, out in d["modes"].items():
        want = np.array([int(v, 16) for v in out], dtype=np.uint32)
        assert_bits_equal(orc.cast(x, mode), want, mode)


def test_survey_spot_values():
    # SURVEY Appendix A "Consequences worth unit-testing explicitly"
    def c(mode, v):
        return float(orc.cast(np.array([v], dtype=np.float32), mode)[0])

    assert c("E4M3_RNE", 0.0087890625) == 0.0087890625
    assert c("E4M3_RNE", 0.0015) == 0.0
    assert c("E4M3_RNE", float("nan")) == 448.0
    assert c("E4M3_RNE", float("-inf")) == -448.0
    assert c("E4M3_RNE", 465.0) == 448.0
    assert c("E3M4_RNE", float("nan")) == 30.0
    assert c("E3M4_RNE", 0.0087890625) == 0.0
    assert c("E5M2_RNE", 65504.0) == float("inf")
    assert c("E5M2_RTZ", 65504.0) == 57344.0
    assert math.isnan(c("E5M2_RNE", float("nan")))
    assert c("E5M2_RNE", -61440.0) == float("-inf")
    assert c("E5M2_RTZ", -61440.0) == -57344.0
    assert c("E5M2_RNE", 1e-5) == 2.0 ** -16
    assert c("E5M2_RNE", 0.51 * 2.0 ** -16) == 2.0 ** -16
    assert c("E5M2_RNE", 0.5 * 2.0 ** -16) == 0.0
    assert c("BFLOAT16_RNE", 65504.0) == 65536.0
    assert c("BFLOAT16_RNE", 465.0) == 464.0
    for mode in DET_MODES + EXT_MODES:
        if "DAZ" in mode:
            continue
        z = orc.cast(np.array([-0.0], dtype=np.float32), mode)
        assert bits(z)[0] == 0x80000000, mode


@pytest.mark.parametrize("mode", EXT_MODES)
def test_rtz_extensions_properties(mode):
    """E4M3_RTZ / E3M4_RTZ / BFLOAT16_RTZ are not defined by the reference ("parity unpinned",
    SURVEY 8(a)): pinned by properties only."""
    g = npz("rawbits.npz")
    x = g["input_bits"].view(np.float32)
    y = orc.cast(x, mode)
    fin = np.isfinite(x) & np.isfinite(y)
    # never away from zero -- measured from the fp16-rounded input for the FP8 formats, whose
    # first step is an RNE convert to fp16 (cpp/fpemu_impl.cpp:1067) that may itself round up
    with np.errstate(over="ignore"):
        x0 = x if mode.startswith("BFLOAT16") else x.astype(np.float16).astype(np.float32)
    assert np.all(np.abs(y[fin]) <= np.abs(x0[fin]))
    assert_bits_equal(orc.cast(y, mode), y, mode + " idempotent")   # idempotent (NaN -> max is a fixpoint too)
    rne = orc.cast(x, mode.replace("RTZ", "RNE"))
    exact = fin & (rne == x)
    assert_bits_equal(y[exact], x[exact], mode + " exact inputs")


def _sr_counts(mode, x_bits, lo_bits, hi_bits, n, seed):
    x = np.full(n, np.array([x_bits], dtype=np.uint32).view(np.float32)[0], dtype=np.float32)
    y = bits(orc.cast(x, mode, seed=seed, offset=7))
    return int((y == lo_bits).sum()), int((y == hi_bits).sum())


def test_sr_distribution_matches_reference():
    """Round-up probability of the oracle's Philox-driven SR == the scalar reference's, case by
    case, within 5 sigma of the binomial noise of both samples; and == residue / 2^k exactly in
    expectation."""
    d = jload("sr_hist.json")
    n = d["N"]
    k_of = {"E5M2_STOCHASTIC": 8, "E4M3_STOCHASTIC": 7, "E3M4_STOCHASTIC": 6,
            "BFLOAT16_STOCHASTIC": 16, "FLOAT16_STOCHASTIC": 13}
    for i, c in enumerate(d["cases"]):
        lo, hi, xb = int(c["lo_bits"], 16), int(c["hi_bits"], 16), int(c["x_bits"], 16)
        n_lo, n_hi = _sr_counts(c["mode"], xb, lo, hi, n, seed=100 + i)
        assert n_lo + n_hi == n, c
        p_ref = c["n_hi"] / n
        p_got = n_hi / n
        sigma = math.sqrt(max(p_ref * (1 - p_ref), 1.0 / n) / n) * math.sqrt(2.0)
        assert abs(p_got - p_ref) <= 5 * sigma + 2.0 / n, (c, p_got, p_ref)
        if c["mode"] in k_of and xb >> 23 == 0x7F:  # x in [1, 2): residue is readable off the bits
            k = k_of[c["mode"]]
            drop = {8: 13 + 8, 7: 13 + 7, 6: 13 + 6, 16: 16, 13: 13}[k]
            r = (xb & ((1 << drop) - 1)) >> (drop - k)
            p = r / float(1 << k)
            s = math.sqrt(max(p * (1 - p), 1.0 / n) / n)
            assert abs(p_got - p) <= 5 * s + 1.0 / n, (c, p_got, p)


def test_sr_exact_values_never_move_and_offsets_differ():
    x = np.array([1.0, -2.0, 0.5, 24.0, 0.0, 1.5], dtype=np.float32).repeat(4096)
    for mode in ("E5M2_STOCHASTIC", "E4M3_STOCHASTIC", "E3M4_STOCHASTIC", "BFLOAT16_STOCHASTIC",
                 "FLOAT16_STOCHASTIC"):
        assert_bits_equal(orc.cast(x, mode, seed=1, offset=0), x, mode)
    y = np.full(1 << 14, 1.1, dtype=np.float32)
    a = orc.cast(y, "E5M2_STOCHASTIC", seed=1, offset=0)
    b = orc.cast(y, "E5M2_STOCHASTIC", seed=1, offset=1)
    c = orc.cast(y, "E5M2_STOCHASTIC", seed=2, offset=0)
    assert np.array_equal(a, orc.cast(y, "E5M2_STOCHASTIC", seed=1, offset=0))
    assert not np.array_equal(a, b) and not np.array_equal(a, c)


def test_sr_sharding_invariance():
    """Noise is a function of the GLOBAL element index: casting shards with elem_base equals
    casting the whole buffer."""
    rng = np.random.default_rng(9)
    x = rng.standard_normal(5000).astype(np.float32)
    whole = orc.cast(x, "E4M3_STOCHASTIC", seed=11, offset=3)
    parts = [orc.cast(x[a:b], "E4M3_STOCHASTIC", seed=11, offset=3, elem_base=a)
             for a, b in ((0, 2048), (2048, 3072), (3072, 5000))]
    assert_bits_equal(np.concatenate(parts), whole)


def _restate_quantize_tensor(x, dtype, scheme, scaling, gs):
    """quantize_tensor (qutils.py:264-345) restated on top of the oracle primitives."""
    mode = (dtype + "_" + scheme).upper()
    fmax = FLT_MAX[dtype.upper()]
    if scaling == "None":
        return orc.cast(x, mode)
    if scaling == "per-tensor":
        return orc.cast(x, mode, scale=orc.scale_from_amax(orc.amax(x), fmax))
    if scaling == "per-channel":
        out = np.empty_like(x)
        for c in range(x.shape[0]):
            out[c] = orc.cast(x[c], mode, scale=orc.scale_from_amax(orc.amax(x[c]), fmax))
        return out
    if scaling == "per-channel-mean":
        # qutils.py:294-301: the statistic is torch.mode of the row; scale = flt_min / |mode| computed by torch as
        # reciprocal-then-multiply in fp32, 1.0 when |mode| <= 1e-6 or when the quotient is below 1
        fmin = np.float32(FLT_MIN[dtype.upper()])
        out = np.empty_like(x)
        for c in range(x.shape[0]):
            vals, counts = np.unique(x[c], return_counts=True)
            a = np.abs(np.float32(vals[np.argmax(counts)]))
            s = np.float32(1.0)
            if a > np.float32(1e-6):
                with np.errstate(divide="ignore"):
                    s = np.float32(np.float32(1.0) / a) * fmin
                s = np.float32(1.0) if s < 1.0 else np.float32(s)
            out[c] = orc.cast(x[c], mode, scale=float(s))
        return out
    if scaling == "fine-grained":
        out = np.empty_like(x)
        for k in range(x.shape[0]):
            crs = x[k]
            chunks = max(1, int(crs.shape[0] / gs))
            size = -(-crs.shape[0] // chunks)  # torch.chunk: ceil-sized pieces
            for a in range(0, crs.shape[0], size):
                sub = crs[a:a + size]
                s = orc.scale_from_amax(orc.amax(sub), fmax, clamp=False)
                out[k, a:a + size] = orc.cast(sub, mode, scale=s)
        return out
    raise KeyError(scaling)


def test_quantize_tensor_policies_against_reference_python():
    g = npz("qtensor.npz")
    checked = 0
    for key in g["names"]:
        tname, dtype, scheme, scaling, gs, inplace = str(key).split("|")
        if scaling == "per-tensor-mean":
            continue  # torch.mean based policy: host logic, covered on the GPU side
        x = g["in/" + tname].view(np.float32)
        want = g["out/" + str(key)]
        got = _restate_quantize_tensor(x, dtype, scheme, scaling, int(gs))
        assert_bits_equal(got, want, str(key))
        checked += 1
    assert checked >= 100


def test_scale_rule_edges():
    assert float(orc.scale_from_amax(0.0, 448.0)) == 65500.0          # vmax = 0 -> inf -> 6.55e4
    assert math.isnan(float(orc.scale_from_amax(float("nan"), 448.0)))
    assert float(orc.scale_from_amax(1.0, 448.0)) == 448.0
    assert float(orc.scale_from_amax(448.0 / 32750.0, 448.0)) in (65500.0, 32750.0)
    x = np.array([1.0, float("nan"), -3.0], dtype=np.float32)
    assert math.isnan(float(orc.amax(x)))
    assert float(orc.amax(np.array([1.0, -7.5, 3.0], dtype=np.float32))) == 7.5


@pytest.mark.parametrize("build,fused", [("fma", True), ("nofma", False)])
def test_gemm_oracle_matches_reference_builds(build, fused):
    """orc_gemm_f32 vs the reference's simple_gemm_dev.gemm, both ways the reference compiles
    (oracle/build_ref.py load_cmodel: its own flags contract to FMA; -mno-fma -mno-avx512f does not)."""
    g = npz("gemm.npz")
    n = 0
    for key in g["names"]:
        key = str(key)
        M, N, K, at, bt = map(int, key.split("_"))
        A, B = g["A/" + key].view(np.float32), g["B/" + key].view(np.float32)
        for alpha in (1.0, 1.5):
            got = orc.gemm(A, B, alpha, bool(at), bool(bt), fused=fused)
            assert_bits_equal(got, g["C/%s/%s/%g" % (build, key, alpha)], "%s %s %g" % (build, key, alpha))
            n += 1
        c0 = np.arange(M * N, dtype=np.float32).reshape(M, N) * np.float32(0.25) - np.float32(3.0)
        got = orc.gemm(A, B, 0.75, bool(at), bool(bt), out=c0.copy(), fused=fused)
        assert_bits_equal(got, g["Cacc/%s/%s" % (build, key)], "%s %s acc" % (build, key))
    assert n >= 40


# ------------------------------------------------------------------------------------------
# E4M3_IEEE, FP4_NEAREST, block norm -- tests/golden/ext2.npz
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mode", ["E4M3_IEEE_RNE", "FP4_NEAREST"])
def test_ext2_lut_and_raw(mode):
    g = npz("ext2.npz")
    h = np.arange(65536, dtype=np.uint16).view(np.float16).astype(np.float32)
    assert_bits_equal(orc.cast(h, mode), g["lut/" + mode], mode + " lut", nan_payload_ok=True)
    x = g["input_bits"].view(np.float32)
    for i, sc in enumerate(g["raw_scales"]):
        assert_bits_equal(orc.cast(x, mode, scale=sc), g["raw/%s/%d" % (mode, i)], "%s scale %r" % (mode, sc),
                          nan_payload_ok=True)


def test_ext2_e4m3_ieee_limits():
    """Max 240, NaN / Inf saturate, [2^-10, 2^-9) flushes; the scalar routine's 15.0 bug is not restated."""
    x = np.array([14.0, 15.0, 16.0, 100.0, 239.0, 240.0, 248.0, 1e5, np.inf, -np.inf, np.nan, 2.0 ** -9,
                  2.0 ** -10, 1.5 * 2.0 ** -10], dtype=np.float32)
    want = np.array([14.0, 15.0, 16.0, 96.0, 240.0, 240.0, 240.0, 240.0, 240.0, -240.0, 240.0, 2.0 ** -9, 0.0, 0.0],
                    dtype=np.float32)
    got = orc.cast(x, "E4M3_IEEE_RNE")
    assert np.array_equal(np.abs(got), np.abs(want)) and got[9] == -240.0


def test_ext2_blocknorm():
    g = npz("ext2.npz")
    x = g["input_bits"].view(np.float32)
    for case in g["bn_cases"]:
        mode, bs = str(case).split("/")
        assert_bits_equal(orc.cast_blocknorm(x, mode, int(bs)), g["bn/" + str(case)], "bn " + str(case),
                          nan_payload_ok=True)
    x24 = x[4096:4096 + 6144]
    for mode in ("E5M2_RNE", "E4M3_RNE", "E3M4_RNE", "FP4_NEAREST"):
        assert_bits_equal(orc.cast_blocknorm(x24, mode, 24), g["bn24/" + mode], "bn24 " + mode, nan_payload_ok=True)
    # size % block_size != 0: the reference demotes to the plain cast (cpp:2020-2025)
    assert_bits_equal(orc.cast(x, "E4M3_RNE"), g["bn_demoted/E4M3_RNE/100"], "demoted", nan_payload_ok=True)
    with pytest.raises(ValueError):
        orc.cast_blocknorm(x, "E4M3_RNE", 100)
    with pytest.raises(ValueError):
        orc.cast_blocknorm(x, "BFLOAT16_RNE", 64)


def test_ext2_e4m3_ieee_stochastic_statistics():
    """Oracle (Philox noise) vs the reference's AVX-512 stochastic routine (its own RNG): round-up
    counts agree within binomial noise; pseudo-subnormals round RNE, bit-exact."""
    g = npz("ext2.npz")
    N = int(g["sr/N"][0])
    for xb, lo, hi, n_hi in zip(g["sr/x_bits"], g["sr/lo_bits"], g["sr/hi_bits"], g["sr/n_hi"]):
        x = np.full(N, np.array([xb], dtype=np.uint32).view(np.float32)[0], dtype=np.float32)
        out = bits(orc.cast(x, "E4M3_IEEE_STOCHASTIC", seed=11, offset=int(xb) & 0xFFFF))
        assert np.all((out == lo) | (out == hi))
        ours = int((out == hi).sum())
        p = ((int(xb) >> 13) & 0x7F) / 128.0
        sigma = math.sqrt(max(N * p * (1 - p), 1.0))
        assert abs(ours - N * p) < 5 * sigma, (hex(xb), ours, N * p)
        assert abs(int(n_hi) - N * p) < 5 * sigma, (hex(xb), int(n_hi), N * p)
    sub = g["sr/sub_in"].view(np.float32)
    assert_bits_equal(orc.cast(sub, "E4M3_IEEE_STOCHASTIC", seed=3), g["sr/sub_out"], "sub-range RNE under SR")


@pytest.mark.parametrize("build,fused", [("fma", True), ("nofma", False)])
def test_conv_oracle_matches_reference_builds(build, fused):
    """orc_conv2d_{fp,bp,wu} vs the reference's simple_conv2d_dev run single-threaded, both builds."""
    g = npz("conv.npz")
    for ci, (N, C, H, W, K, R, S, st, pd) in enumerate(g["cases"].tolist()):
        OH, OW = (H + 2 * pd - R) // st + 1, (W + 2 * pd - S) // st + 1
        x = g["%d/x" % ci].view(np.float32).reshape(N, C, H, W)
        w = g["%d/w" % ci].view(np.float32).reshape(K, C, R, S)
        b = g["%d/b" % ci].view(np.float32)
        go = g["%d/go" % ci].view(np.float32).reshape(N, K, OH, OW)
        tag = "%s case %d" % (build, ci)
        assert_bits_equal(orc.conv2d_fp(x, w, b, st, pd, fused), g["%d/%s/fp" % (ci, build)], tag + " fp")
        assert_bits_equal(orc.conv2d_bp(go, w, (H, W), st, pd, fused), g["%d/%s/bp" % (ci, build)], tag + " bp")
        assert_bits_equal(orc.conv2d_wu(go, x, (R, S), st, pd, fused), g["%d/%s/wu" % (ci, build)], tag + " wu")
    with pytest.raises(ValueError):
        orc.conv2d_fp(np.zeros((1, 96, 4, 4), np.float32), np.zeros((8, 96, 1, 1), np.float32))
