"""Generate the golden fixtures in tests/golden/ from the REFERENCE itself.

Run in the build container only (needs /root/reference):

    OMP_NUM_THREADS=1 python3 tests/golden/make_golden.py

The reference's fpemu_cpp extension is compiled from its own sources by oracle/build_ref.py
(native = AVX-512 routines, scalar = -mno-avx512f) and its Python package is imported from
/root/reference.  Only data (inputs and the reference's outputs) is written here; no reference
source text is stored.  SURVEY.md section 8(a) names which build is canonical per mode.

Fixtures (all loadable with numpy.load(allow_pickle=False) / json):
  luts.npz        every fp16 pattern (as fp32 input, scale 1) -> output bits, per deterministic mode
  rawbits.npz     32 Ki raw random fp32 bit patterns + 32 Ki log-normal values -> output bits per mode
  scaled.npz      4 Ki log-normal values x 5 scales x 4 modes
  testpy64.json   the 64 edge-case inputs the reference's own pytquant/test.py holds x modes
  sr_hist.json    scalar-reference stochastic-rounding round-up counts (distribution pin)
  qtensor.npz     reference qutils.quantize_tensor (per-tensor / per-channel / fine-grained / mean)
  models.npz      reference mpt_emu.quantize_model / scale_shift on tiny seeded models
  gemm.npz        reference cmodel simple_gemm_dev.gemm, FMA build and -mno-fma build
  conv.npz        reference cmodel simple_conv2d_dev conv2d_fp / _bp / _wu, FMA build and -mno-fma build
  callers.json    repr of every model_qconfig_dict entry the reference's four emulators build for a module holding
                  every whitelisted type (training / inference / calibration, with and without exempt / fused lists)
  callers.npz     training-mode hooks (HybridEmulator / E5M2Emulator / E4M3Emulator(train=True), optimizer=None) on a
                  tiny MLP: hooked weights and layer inputs after a forward pass; a 2-layer tiny BERT built with the
                  module_wrappers through mpt_emu.quantize_model(dtype="e4m3" / "hybrid"): weights, hook inputs, output
  ext2.npz        E4M3_IEEE_RNE (AVX-512 routine = canonical), FP4_NEAREST, block_norm=True outputs,
                  E4M3_IEEE_STOCHASTIC round-up counts
"""
import json
import os
import re
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import build_ref  # noqa: E402

torch.set_num_threads(1)  # the reference's scalar SR shares RNG state across OpenMP threads

NAT = build_ref.load("native")
SCA = build_ref.load("scalar")

# mode -> build that is canonical for it; RNE of the FP8 formats agree on both.
DET_MODES = {
    "E5M2_RNE": NAT, "E4M3_RNE": NAT, "E3M4_RNE": NAT,
    "E5M2_RTZ": SCA, "E5M2_RNAZ": SCA, "E5M2_RNTZ": SCA, "E5M2_RPINF": SCA, "E5M2_RNINF": SCA,
    "E5M2_DAZ_RNE": SCA, "E5M2_DAZ_RNAZ": SCA, "E5M2_DAZ_RNTZ": SCA,
    "BFLOAT16_RNE": NAT, "FLOAT16_RNE": SCA, "FLOAT16_DAZ_RNE": SCA,
}


def ref_cast(mod, x, mode, scale=1.0):
    t = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32).copy())
    out = mod.forward(t, mode, t.numel(), False, float(scale), False, 1)[0]
    return out.numpy().view(np.uint32).copy()


def lognormal(rng, n):
    return (rng.standard_normal(n) * np.exp(rng.standard_normal(n) * 4.0)).astype(np.float32)


def gen_luts():
    h = np.arange(65536, dtype=np.uint16)
    x = h.view(np.float16).astype(np.float32)
    out = {"input_bits": x.view(np.uint32)}
    for mode, mod in DET_MODES.items():
        if mode.startswith("BFLOAT16"):
            continue
        out[mode] = ref_cast(mod, x, mode)
    np.savez_compressed(os.path.join(HERE, "luts.npz"), **out)


def gen_rawbits():
    rng = np.random.default_rng(20241220)
    raw = rng.integers(0, 2 ** 32, size=32768, dtype=np.uint64).astype(np.uint32)
    ln = lognormal(rng, 32768).view(np.uint32)
    x = np.concatenate([raw, ln])
    out = {"input_bits": x}
    for mode, mod in DET_MODES.items():
        out[mode] = ref_cast(mod, x.view(np.float32), mode)
    np.savez_compressed(os.path.join(HERE, "rawbits.npz"), **out)


SCALES = [0.37, 3.0, 116.894, 0.0078125, 6.55e4]


def gen_scaled():
    rng = np.random.default_rng(77)
    x = lognormal(rng, 4096)
    out = {"input_bits": x.view(np.uint32), "scales": np.array(SCALES, dtype=np.float32)}
    for mode in ("E5M2_RNE", "E4M3_RNE", "E3M4_RNE", "E5M2_RTZ"):
        for i, s in enumerate(SCALES):
            out["%s/%d" % (mode, i)] = ref_cast(DET_MODES[mode], x, mode, np.float32(s))
    np.savez_compressed(os.path.join(HERE, "scaled.npz"), **out)


def gen_testpy64():
    # The only reference-authored known-input set: the 64-value tensor literal in the
    # reference's pytquant/test.py.  Values are parsed out of that file at generation time.
    src = open("/root/reference/mpemu/pytquant/test.py").read()
    vals = None
    for body in re.findall(r"\ninput = torch\.tensor\(np\.array\(\[\[(.*?)\]\]\)", src, re.S):
        v = [float(t) for t in re.findall(r"[-+]?\d+\.?\d*(?:[eE][-+]?\d+)?", body)]
        if len(v) == 64:  # the active literal (the other one is the commented-out FP4 vector)
            vals = v
    assert vals is not None
    x = np.array(vals, dtype=np.float32)
    doc = {"input_bits": ["%08x" % v for v in x.view(np.uint32)], "modes": {}}
    for mode, mod in DET_MODES.items():
        # the reference's scalar routines handle a 64-element call's (size % 32 == 0) through the
        # AVX-512 prefix on the native build; DET_MODES already maps each mode to its canonical build
        doc["modes"][mode] = ["%08x" % v for v in ref_cast(mod, x, mode)]
    with open(os.path.join(HERE, "testpy64.json"), "w") as f:
        json.dump(doc, f, indent=0)


def gen_sr_hist():
    """Round-up counts of the scalar reference's stochastic modes.

    For each format: N draws of a constant input whose dropped field has residue r; count how
    many outputs equal the upper neighbour.  The scalar RNG is stateful and unseeded, so these are
    statistical pins (binomial), not bit patterns."""
    N = 1 << 16
    doc = {"N": N, "cases": []}

    def run(mode, xs, lo_hi):
        for x, (lo, hi), p in zip(xs, lo_hi, [None] * len(xs)):
            out = ref_cast(SCA, np.full(N, x, dtype=np.float32), mode).view(np.float32)
            n_hi = int((out == np.float32(hi)).sum())
            n_lo = int((out == np.float32(lo)).sum())
            doc["cases"].append({"mode": mode, "x_bits": "%08x" % np.float32(x).view(np.uint32),
                                 "lo_bits": "%08x" % np.float32(lo).view(np.uint32),
                                 "hi_bits": "%08x" % np.float32(hi).view(np.uint32),
                                 "n_lo": n_lo, "n_hi": n_hi})

    # FP8 formats: x = fp16 pattern with exponent 0 (1.0 <= x < 2) and every residue of the
    # dropped field at a fixed kept mantissa; k = 8 / 7 / 6 dropped bits.
    for mode, k in (("E5M2_STOCHASTIC", 8), ("E4M3_STOCHASTIC", 7), ("E3M4_STOCHASTIC", 6)):
        step = max(1, (1 << k) // 32)
        rs = list(range(0, 1 << k, step))
        base = 0x3C00 | (1 << k)  # kept mantissa LSB set
        xs, lh = [], []
        for r in rs:
            h = np.array([base | r], dtype=np.uint16).view(np.float16).astype(np.float32)[0]
            lo = np.array([base], dtype=np.uint16).view(np.float16).astype(np.float32)[0]
            hi = np.array([base + (1 << k)], dtype=np.uint16).view(np.float16).astype(np.float32)[0]
            xs.append(h)
            lh.append((lo, hi))
        run(mode, xs, lh)
    # E4M3 / E3M4 pseudo-subnormal range (fp16-normal, so SR applies after the dshift clear)
    run("E4M3_STOCHASTIC", [0.005], [(0.0048828125, 0.00537109375)])
    # BF16: 16 dropped bits of the fp32 pattern
    for r in (0x0000, 0x0001, 0x1000, 0x4000, 0x8000, 0xC000, 0xF000, 0xFFFF):
        u = np.array([0x3F810000 | r], dtype=np.uint32)
        x = u.view(np.float32)[0]
        lo = np.array([0x3F810000], dtype=np.uint32).view(np.float32)[0]
        hi = np.array([0x3F820000], dtype=np.uint32).view(np.float32)[0]
        run("BFLOAT16_STOCHASTIC", [x], [(lo, hi)])
    # FP16: 13 dropped bits of the fp32 pattern
    for r in (0x0000, 0x0001, 0x0400, 0x1000, 0x1800, 0x1FFF):
        u = np.array([0x3F802000 | r], dtype=np.uint32)
        x = u.view(np.float32)[0]
        lo = np.array([0x3F802000], dtype=np.uint32).view(np.float32)[0]
        hi = np.array([0x3F804000], dtype=np.uint32).view(np.float32)[0]
        run("FLOAT16_STOCHASTIC", [x], [(lo, hi)])
    # the survey's spot values
    run("E5M2_STOCHASTIC", [1.1], [(1.0, 1.25)])
    run("E4M3_STOCHASTIC", [1.03], [(1.0, 1.125)])
    run("E3M4_STOCHASTIC", [1.01], [(1.0, 1.0625)])
    with open(os.path.join(HERE, "sr_hist.json"), "w") as f:
        json.dump(doc, f, indent=0)


def gen_qtensor():
    """Reference qutils.quantize_tensor outputs (the Python caller of the cast, qutils.py:264-345)."""
    # scalar build: the AVX-512 prefix of the native build silently runs RNE for E5M2_RTZ
    # (SURVEY 8(a) row a2); every RNE case is identical on both builds.
    build_ref.import_reference_package("scalar")
    from mpemu import qutils

    g = torch.Generator().manual_seed(5)
    tensors = {
        "randn": torch.randn(24, 40, generator=g),
        "small": torch.randn(24, 40, generator=g) * 1e-3,
        "large": torch.randn(24, 40, generator=g) * 300.0,
        "tiny": torch.randn(24, 40, generator=g) * 1e-7,
        "zeros": torch.zeros(8, 16),
        "conv": torch.randn(6, 4, 3, 3, generator=g) * 0.2,
        "odd": torch.randn(7, 13, generator=g) * 5.0,
    }
    # rows with ONE most frequent value each (torch.mode is what the per-channel "mean" policy reads,
    # qutils.py:297; with all-distinct values its result is a tie-break that differs between devices)
    modal = torch.randn(24, 40, generator=g)
    for r, v in enumerate([0.0, 1e-7, -1e-7, 3e-6, 1e-5, -2e-5, 1e-4, 1.5258789e-05, 0.5, -0.003, 7.0, 1e-3,
                           float("inf"), 2e-6, 9e-7, 1.1e-6, -4e-4, 6e-5, 3e-5, 1.0, 5e-6, -8e-6, 2.5e-4, 1e-6]):
        modal[r, r % 7:r % 7 + 12] = v
    tensors["modal"] = modal
    cases = [
        ("e4m3", "rne", "per-tensor", 1), ("e5m2", "rne", "per-tensor", 1),
        ("e3m4", "rne", "per-tensor", 1), ("e4m3", "rne", "per-channel", 1),
        ("e5m2", "rne", "per-channel", 1), ("e3m4", "rne", "per-channel", 1),
        ("e4m3", "rne", "per-tensor-mean", 1), ("e4m3", "rne", "None", 1),
        ("e5m2", "rtz", "None", 1), ("e4m3", "rne", "fine-grained", 8),
        ("e3m4", "rne", "fine-grained", 4),
        ("e5m2", "rne", "per-channel-mean", 1), ("e4m3", "rne", "per-channel-mean", 1),
        ("e5m2", "rne", "per-tensor-mean", 1),
    ]
    out = {}
    names = []
    for tname, t in tensors.items():
        out["in/" + tname] = t.numpy().view(np.uint32)
        for dtype, scheme, scaling, gs in cases:
            if scaling == "fine-grained" and (t.dim() < 2 or tname in ("zeros", "modal")):  # all-zero group: inf scale, exit()
                continue
            if scaling == "per-channel-mean" and tname not in ("modal", "zeros"):
                continue
            cfg = qutils.TensorQuantConfig(dtype, scheme, scaling, gs)
            for inplace in (False, True):
                q = qutils.quantize_tensor(t.clone(), cfg, None, inplace=inplace)
                key = "%s|%s|%s|%s|%d|%d" % (tname, dtype, scheme, scaling, gs, int(inplace))
                out["out/" + key] = q.detach().numpy().view(np.uint32)
                names.append(key)
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "qtensor.npz"), **out)


def gen_gemm():
    """Reference cmodel GEMM (simple_gemm_dev.gemm) from both builds: "fma" = the reference's own flags
    on this FMA host (vfmadd231ps chain), "nofma" = -mno-fma -mno-avx512f (separate multiply and add)."""
    builds = {"fma": build_ref.load_cmodel("simple_gemm_dev"),
              "nofma": build_ref.load_cmodel("simple_gemm_dev", nofma=True)}
    rng = np.random.default_rng(4242)
    out = {}
    names = []
    for (M, N, K) in [(16, 16, 64), (33, 17, 70), (96, 80, 200), (70, 133, 100), (5, 3, 1), (32, 48, 1024)]:
        for at in (False, True):
            for bt in (False, True):
                if (M * N > 9000 or K > 512) and at != bt:
                    continue
                A = (rng.standard_normal((K, M) if at else (M, K)) * np.exp(rng.standard_normal() * 2)).astype(np.float32)
                B = rng.standard_normal((N, K) if bt else (K, N)).astype(np.float32)
                key = "%d_%d_%d_%d_%d" % (M, N, K, int(at), int(bt))
                out["A/" + key] = A.view(np.uint32)
                out["B/" + key] = B.view(np.uint32)
                names.append(key)
                for bname, g in builds.items():
                    for alpha in (1.0, 1.5):
                        C = torch.zeros(M, N)
                        g.gemm(C, torch.from_numpy(A.copy()), torch.from_numpy(B.copy()), alpha, at, bt)
                        out["C/%s/%s/%g" % (bname, key, alpha)] = C.numpy().view(np.uint32)
                    # accumulate into a non-zero output: C += alpha*acc + 0*C (simple_gemm_impl.cpp:226-231)
                    C = torch.from_numpy(np.arange(M * N, dtype=np.float32).reshape(M, N) * 0.25 - 3.0)
                    g.gemm(C, torch.from_numpy(A.copy()), torch.from_numpy(B.copy()), 0.75, at, bt)
                    out["Cacc/%s/%s" % (bname, key)] = C.numpy().view(np.uint32)
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "gemm.npz"), **out)


CONV_CASES = [  # N, C, H, W, K, R, S, stride, pad
    (1, 3, 9, 9, 8, 3, 3, 1, 1), (1, 16, 8, 8, 32, 3, 3, 2, 1), (2, 64, 7, 7, 64, 1, 1, 1, 0),
    (2, 128, 6, 6, 64, 3, 3, 1, 1), (1, 64, 6, 6, 128, 3, 3, 2, 1), (1, 3, 15, 15, 16, 7, 7, 2, 3),
    (1, 5, 9, 7, 7, 3, 3, 1, 0), (1, 8, 10, 10, 8, 2, 2, 2, 0), (1, 4, 11, 11, 4, 5, 5, 3, 2),
    (3, 64, 5, 5, 64, 3, 3, 1, 1),
]


def gen_conv():
    """The reference is run with ONE thread: its OpenMP loops over channel blocks / rows update shared
    results without synchronisation.  Cases keep to shapes it handles (C, K < 64 with N = 1 -- its packing
    collapses the batch otherwise -- or multiples of 64)."""
    builds = {"fma": build_ref.load_cmodel("simple_conv2d_dev"),
              "nofma": build_ref.load_cmodel("simple_conv2d_dev", nofma=True)}
    rng = np.random.default_rng(99)
    out = {"cases": np.array(CONV_CASES, dtype=np.int64)}
    for ci, (N, C, H, W, K, R, S, st, pd) in enumerate(CONV_CASES):
        OH, OW = (H + 2 * pd - R) // st + 1, (W + 2 * pd - S) // st + 1
        x = rng.standard_normal((N, C, H, W)).astype(np.float32)
        w = (rng.standard_normal((K, C, R, S)) * 0.2).astype(np.float32)
        b = rng.standard_normal(K).astype(np.float32)
        go = rng.standard_normal((N, K, OH, OW)).astype(np.float32)
        if ci == 0:
            b[1] = -0.0
            x[0, :, 2, 2] = 0.0
        for nm, a in (("x", x), ("w", w), ("b", b), ("go", go)):
            out["%d/%s" % (ci, nm)] = a.view(np.uint32)
        for bname, m in builds.items():
            y = torch.empty(N, K, OH, OW)
            m.conv2d_fp(y, torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b), st, pd, 1, 1)
            gi = torch.zeros(N, C, H, W)
            m.conv2d_bp(gi, torch.from_numpy(go), torch.from_numpy(w), st, pd, 1, 1)
            gw = torch.zeros(K, C, R, S)
            m.conv2d_wu(gw, torch.from_numpy(go), torch.from_numpy(x), st, pd, 1, 1)
            out["%d/%s/fp" % (ci, bname)] = y.numpy().view(np.uint32)
            out["%d/%s/bp" % (ci, bname)] = gi.numpy().view(np.uint32)
            out["%d/%s/wu" % (ci, bname)] = gw.numpy().view(np.uint32)
    np.savez_compressed(os.path.join(HERE, "conv.npz"), **out)


def ref_blocknorm(mod, x, mode, bs):
    t = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32).copy())
    return mod.forward(t, mode, t.numel(), False, 1.0, True, int(bs))[0].numpy().view(np.uint32).copy()


def gen_ext2():
    out = {}
    h = np.arange(65536, dtype=np.uint16).view(np.float16).astype(np.float32)
    out["lut/E4M3_IEEE_RNE"] = ref_cast(NAT, h, "E4M3_IEEE_RNE")
    out["lut/FP4_NEAREST"] = ref_cast(SCA, h, "FP4_NEAREST")
    rng = np.random.default_rng(424242)
    raw = rng.integers(0, 2 ** 32, size=8192, dtype=np.uint64).astype(np.uint32).view(np.float32)
    x = np.concatenate([raw, lognormal(rng, 8192)])
    x[8192 + 128:8192 + 192] = 0.0          # all-zero blocks (scale 0 -> 0 * inf)
    x[8192 + 300] = 0.0
    x[8192 + 301] = -0.0
    x[8192 + 512:8192 + 576] = np.float32(1e-41) * np.arange(64, dtype=np.float32)  # denormal maxima
    out["input_bits"] = x.view(np.uint32)
    for i, sc in enumerate((1.0, 0.37, 37.5)):
        out["raw/E4M3_IEEE_RNE/%d" % i] = ref_cast(NAT, x, "E4M3_IEEE_RNE", np.float32(sc))
        out["raw/FP4_NEAREST/%d" % i] = ref_cast(SCA, x, "FP4_NEAREST", np.float32(sc))
    out["raw_scales"] = np.array([1.0, 0.37, 37.5], dtype=np.float32)
    # block_norm=True.  E4M3_IEEE only with block sizes % 32 == 0 on the native build (other sizes run
    # its scalar routine, whose saturation test is wrong); the rest on the scalar build.
    cases = []
    for mode, mod, sizes in (("E5M2_RNE", SCA, (1, 8, 32, 64, 4096, 16384)), ("E5M2_RTZ", SCA, (8, 64)),
                             ("E5M2_RNAZ", SCA, (64,)), ("E4M3_RNE", SCA, (1, 8, 32, 64, 4096, 16384)),
                             ("E3M4_RNE", SCA, (1, 8, 32, 64, 4096, 16384)),
                             ("FP4_NEAREST", SCA, (8, 32, 64, 4096, 16384)),
                             ("E4M3_IEEE_RNE", NAT, (32, 64, 4096, 16384))):
        for bs in sizes:
            out["bn/%s/%d" % (mode, bs)] = ref_blocknorm(mod, x, mode, bs)
            cases.append("%s/%d" % (mode, bs))
    x24 = x[4096:4096 + 6144]
    for mode in ("E5M2_RNE", "E4M3_RNE", "E3M4_RNE", "FP4_NEAREST"):
        out["bn24/%s" % mode] = ref_blocknorm(SCA, x24, mode, 24)
    out["bn_cases"] = np.array(cases)
    # a block_size that does not divide size: the reference demotes block_norm (cpp:2020-2025)
    out["bn_demoted/E4M3_RNE/100"] = ref_blocknorm(SCA, x, "E4M3_RNE", 100)
    # E4M3_IEEE_STOCHASTIC round-up counts, AVX-512 routine (N % 32 == 0)
    N = 1 << 16
    k = 7
    xb, lob, hib, nlo, nhi = [], [], [], [], []
    base = 0x3C00 | (1 << k)
    for r in range(0, 1 << k, 4):
        xv = np.array([base | r], dtype=np.uint16).view(np.float16).astype(np.float32)[0]
        lo = np.array([base], dtype=np.uint16).view(np.float16).astype(np.float32)[0]
        hi = np.array([base + (1 << k)], dtype=np.uint16).view(np.float16).astype(np.float32)[0]
        o = ref_cast(NAT, np.full(N, xv, dtype=np.float32), "E4M3_IEEE_STOCHASTIC").view(np.float32)
        xb.append(np.float32(xv).view(np.uint32)); lob.append(np.float32(lo).view(np.uint32))
        hib.append(np.float32(hi).view(np.uint32))
        nlo.append(int((o == lo).sum())); nhi.append(int((o == hi).sum()))
    # pseudo-subnormal range rounds RNE under STOCHASTIC (cpp:1314): deterministic
    sub = np.array([0.005, 0.0051, 0.0107, 0.0137, 0.0029296875, 0.0021], dtype=np.float32)
    out["sr/sub_in"] = sub.view(np.uint32)
    out["sr/sub_out"] = ref_cast(NAT, np.tile(sub, 32)[:192], "E4M3_IEEE_STOCHASTIC")[:6]
    out["sr/x_bits"] = np.array(xb, dtype=np.uint32)
    out["sr/lo_bits"] = np.array(lob, dtype=np.uint32)
    out["sr/hi_bits"] = np.array(hib, dtype=np.uint32)
    out["sr/n_lo"] = np.array(nlo, dtype=np.int64)
    out["sr/n_hi"] = np.array(nhi, dtype=np.int64)
    out["sr/N"] = np.array([N], dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, "ext2.npz"), **out)


def _tiny_models():
    torch.manual_seed(1234)
    mlp = torch.nn.Sequential(torch.nn.Linear(40, 64), torch.nn.ReLU(), torch.nn.Linear(64, 24))
    conv = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.BatchNorm2d(8), torch.nn.ReLU(),
                               torch.nn.Conv2d(8, 6, 3))
    with torch.no_grad():
        conv[1].running_mean.copy_(torch.randn(8) * 0.3)
        conv[1].running_var.copy_(torch.rand(8) + 0.5)
        conv[1].weight.copy_(torch.randn(8))
        conv[1].bias.copy_(torch.randn(8))
    x_mlp = torch.randn(16, 40) * 2.0
    x_conv = torch.randn(4, 3, 10, 10)
    return mlp, conv, x_mlp, x_conv


def gen_models():
    """Reference mpt_emu.quantize_model / scale_shift on tiny seeded models (CPU).  Weights after
    quantisation and the tensors entering each hooked module are exact pins; layer outputs go
    through the host BLAS and are compared with a tolerance on the GPU side."""
    import copy

    build_ref.import_reference_package("scalar")
    from mpemu import mpt_emu, scale_shift

    mlp, conv, x_mlp, x_conv = _tiny_models()
    out = {"x_mlp": x_mlp.numpy().view(np.uint32), "x_conv": x_conv.numpy().view(np.uint32)}
    for k, v in mlp.state_dict().items():
        out["mlp_init/" + k] = v.numpy().view(np.uint32)
    for k, v in conv.state_dict().items():
        if v.dtype == torch.float32:
            out["conv_init/" + k] = v.numpy().view(np.uint32)
    for dtype in ("e4m3", "e5m2", "e3m4", "hybrid"):
        for calibrate in (False, True):
            m = copy.deepcopy(mlp).eval()
            m, emu = mpt_emu.quantize_model(m, dtype=dtype, calibrate=calibrate, device="cpu")
            seen = {}

            def grab(mod, inp, tag):
                seen[tag] = inp[0].detach().clone()

            h0 = m[0].register_forward_hook(lambda mod, i, o: grab(mod, i, "in0"))
            h2 = m[2].register_forward_hook(lambda mod, i, o: grab(mod, i, "in2"))
            with torch.no_grad():
                y = m(x_mlp.clone())
            tag = "mlp/%s/%d/" % (dtype, int(calibrate))
            out[tag + "w0"] = m[0].weight.detach().numpy().view(np.uint32)
            out[tag + "w2"] = m[2].weight.detach().numpy().view(np.uint32)
            out[tag + "in0"] = seen["in0"].numpy().view(np.uint32)
            out[tag + "in2"] = seen["in2"].numpy().view(np.uint32)
            out[tag + "y"] = y.numpy().view(np.uint32)
            h0.remove(); h2.remove()
    c = copy.deepcopy(conv).eval()
    with torch.no_grad():
        y_bn = c(x_conv)
    c2 = scale_shift.replace_batchnorms_with_scaleshifts(copy.deepcopy(conv).eval())
    with torch.no_grad():
        y_ss = c2(x_conv)
    out["conv/ss_weight"] = c2[1].weight.detach().numpy().view(np.uint32)
    out["conv/ss_bias"] = c2[1].bias.detach().numpy().view(np.uint32)
    out["conv/y_bn"] = y_bn.numpy().view(np.uint32)
    out["conv/y_ss"] = y_ss.numpy().view(np.uint32)
    c3, emu = mpt_emu.quantize_model(c2, dtype="e4m3", device="cpu")
    with torch.no_grad():
        y_q = c3(x_conv)
    out["conv/e4m3/w0"] = c3[0].weight.detach().numpy().view(np.uint32)
    out["conv/e4m3/w3"] = c3[3].weight.detach().numpy().view(np.uint32)
    out["conv/e4m3/y"] = y_q.numpy().view(np.uint32)
    np.savez_compressed(os.path.join(HERE, "models.npz"), **out)


def gen_callers():
    """The callers of the cast, pinned against the reference's own Python (CPU): which module gets which
    config (reprs), and the tensors the training-mode and wrapper-module hooks produce."""
    import copy

    build_ref.import_reference_package("scalar")
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _zoo
    from mpemu import mpt_emu
    from mpemu.e3m4_emu import E3M4Emulator
    from mpemu.e4m3_emu import E4M3Emulator
    from mpemu.e5m2_emu import E5M2Emulator
    from mpemu.hybrid_emu import HybridEmulator

    classes = {"e5m2": E5M2Emulator, "e4m3": E4M3Emulator, "e3m4": E3M4Emulator, "hybrid": HybridEmulator}
    cfgs = {}
    torch.manual_seed(99)
    zoo0 = _zoo.Zoo()
    for cname, cls in classes.items():
        for phase in ("training", "inference", "calibrate"):
            for lists in ("none", "lists"):
                zoo = copy.deepcopy(zoo0)
                zoo.train(phase == "training")
                if cname == "e4m3":
                    emu = cls(zoo, None, device="cpu", train=(phase == "training"))
                else:
                    emu = cls(zoo, None, device="cpu")
                if phase == "inference":
                    emu.is_training = False
                    emu.set_default_inference_qconfig()
                elif phase == "calibrate":
                    emu.is_training = False
                    emu.set_calibration_qconfig()
                exempt, fused = (None, None) if lists == "none" else (["lin2", "block.0"], ["lin1", "mm", "lin2"])
                key = "%s/%s/%s" % (cname, phase, lists)
                try:
                    emu.prepare_model(zoo, exempt, fused)
                    cfgs[key] = {"is_training": bool(emu.is_training),
                                 "modules": {k: repr(v) for k, v in emu.model_qconfig_dict.items()},
                                 "unpatched": list(emu.list_unpatched)}
                except Exception as e:  # the reference's own failure modes are part of the record
                    cfgs[key] = {"error": type(e).__name__}
    json.dump(cfgs, open(os.path.join(HERE, "callers.json"), "w"), indent=1, sort_keys=True)

    out = {}
    x = torch.randn(20, 24, generator=torch.Generator().manual_seed(8)) * 1.5
    out["mlp_x"] = x.numpy().view(np.uint32)
    mlp0 = _zoo.tiny_mlp()
    for k, v in mlp0.state_dict().items():
        out["mlp_init/" + k] = v.numpy().view(np.uint32)
    for cname in ("hybrid", "e5m2", "e4m3"):
        m = copy.deepcopy(mlp0).train()
        emu = classes[cname](m, None, device="cpu", train=True) if cname == "e4m3" else classes[cname](m, None, device="cpu")
        emu.prepare_model(m, None, None)
        seen = {}
        hs = [m[i].register_forward_hook(lambda mod, inp, o, i=i: seen.__setitem__(i, (inp[0].detach().clone(), o.detach().clone())))
              for i in (0, 2)]
        y = m(x.clone())
        tag = "mlp_train/%s/" % cname
        out[tag + "w0"] = m[0].weight.detach().numpy().view(np.uint32)
        out[tag + "w2"] = m[2].weight.detach().numpy().view(np.uint32)
        out[tag + "in0"] = seen[0][0].numpy().view(np.uint32)
        out[tag + "in2"] = seen[2][0].numpy().view(np.uint32)
        out[tag + "out0"] = seen[0][1].numpy().view(np.uint32)
        out[tag + "y"] = y.detach().numpy().view(np.uint32)
        for h in hs:
            h.remove()

    bert0, ids = _zoo.tiny_bert()
    out["bert_ids"] = ids.numpy().astype(np.int64)
    for k, v in bert0.state_dict().items():
        out["bert_init/" + k] = v.numpy().view(np.uint32)
    bcfg = {}
    for dtype in ("e4m3", "hybrid"):
        m = copy.deepcopy(bert0).eval()
        m, mpt = mpt_emu.quantize_model(m, dtype=dtype, device="cpu",
                                        list_layers_output_fused=None, list_exempt_layers=None)
        bcfg[dtype] = {k: repr(v) for k, v in mpt.emulator.model_qconfig_dict.items()}
        seen = {}
        names = ["layers.0.q", "layers.0.scores", "layers.0.context", "layers.0.ff1", "layers.1.res2", "pool"]
        mods = dict(m.named_modules())
        hs = [mods[n].register_forward_hook(lambda mod, inp, o, n=n: seen.__setitem__(n, [t.detach().clone() for t in inp if torch.is_tensor(t)]))
              for n in names]
        with torch.no_grad():
            y = m(ids)
        tag = "bert/%s/" % dtype
        for k, v in m.state_dict().items():
            out[tag + "w/" + k] = v.numpy().view(np.uint32)
        for n in names:
            for j, t in enumerate(seen[n]):
                out[tag + "in/%s/%d" % (n, j)] = t.contiguous().numpy().view(np.uint32)
        out[tag + "y"] = y.numpy().view(np.uint32)
        for h in hs:
            h.remove()
    cfgs_b = json.load(open(os.path.join(HERE, "callers.json")))
    cfgs_b["tiny_bert"] = bcfg
    json.dump(cfgs_b, open(os.path.join(HERE, "callers.json"), "w"), indent=1, sort_keys=True)
    np.savez_compressed(os.path.join(HERE, "callers.npz"), **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["luts", "rawbits", "scaled", "testpy64", "sr_hist", "qtensor", "models", "gemm", "ext2", "conv", "callers"]
    for w in which:
        globals()["gen_" + w]()
        print("golden:", w, "done")
