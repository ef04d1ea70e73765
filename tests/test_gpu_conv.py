"""cmodel direct convolution on the GPU vs the reference's simple_conv2d_dev (both builds, run
single-threaded; tests/golden/conv.npz) and vs the oracle's serial accumulation chains.  Bit-exact."""
import numpy as np
import pytest
import torch

import orc
from _util import assert_bits_equal, npz

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
VARIANTS = [("fma", 1, True), ("nofma", 0, False)]


@pytest.fixture(scope="module")
def hip():
    from mpemu.pytquant.hip import _binding as b

    b.lib()
    return b


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def _run_all(hip, x, w, b, go, st, pd, variant):
    N, C, H, W = x.shape
    K, _, R, S = w.shape
    y = torch.empty(go.shape, device=DEV)
    hip.conv2d_fp_(y, _t(x), _t(w), None if b is None else _t(b), st, pd, variant)
    gi = torch.full(x.shape, 7.0, device=DEV)    # overwritten, not accumulated
    hip.conv2d_bp_(gi, _t(go), _t(w), st, pd, variant)
    gw = torch.full(w.shape, 7.0, device=DEV)
    hip.conv2d_wu_(gw, _t(go), _t(x), st, pd, variant)
    return y.cpu().numpy(), gi.cpu().numpy(), gw.cpu().numpy()


@pytest.mark.parametrize("bname,variant,fused", VARIANTS)
def test_conv_vs_reference_fixture(hip, bname, variant, fused):
    g = npz("conv.npz")
    for ci, (N, C, H, W, K, R, S, st, pd) in enumerate(g["cases"].tolist()):
        OH, OW = (H + 2 * pd - R) // st + 1, (W + 2 * pd - S) // st + 1
        x = g["%d/x" % ci].view(np.float32).reshape(N, C, H, W)
        w = g["%d/w" % ci].view(np.float32).reshape(K, C, R, S)
        b = g["%d/b" % ci].view(np.float32)
        go = g["%d/go" % ci].view(np.float32).reshape(N, K, OH, OW)
        y, gi, gw = _run_all(hip, x, w, b, go, st, pd, variant)
        tag = "%s case %d" % (bname, ci)
        assert_bits_equal(y, g["%d/%s/fp" % (ci, bname)], tag + " fp")
        assert_bits_equal(gi, g["%d/%s/bp" % (ci, bname)], tag + " bp")
        assert_bits_equal(gw, g["%d/%s/wu" % (ci, bname)], tag + " wu")


@pytest.mark.parametrize("bname,variant,fused", VARIANTS)
@pytest.mark.parametrize("case", [(2, 3, 32, 32, 64, 7, 7, 2, 3), (4, 64, 14, 14, 64, 3, 3, 1, 1),
                                  (3, 16, 9, 11, 24, 3, 3, 1, 1), (2, 128, 8, 8, 128, 1, 1, 2, 0),
                                  (1, 64, 28, 28, 192, 3, 3, 2, 1), (5, 1, 5, 5, 1, 5, 5, 1, 0),
                                  (40, 64, 28, 28, 256, 3, 3, 1, 1),    # >= 512 tiles: the 128 x 128 kernels
                                  # the implicit-GEMM path (stride 1, 64-channel blocks): 1 x 1, 5 x 5, no padding
                                  # (bp pads by R-1), rectangular images and filters, several channel blocks
                                  (3, 128, 9, 13, 64, 1, 1, 1, 0), (2, 64, 12, 10, 128, 5, 5, 1, 2),
                                  (2, 64, 10, 10, 64, 3, 3, 1, 0), (2, 64, 8, 8, 64, 3, 1, 1, 1),
                                  (2, 192, 7, 7, 128, 3, 3, 1, 1), (33, 128, 56, 56, 128, 3, 3, 1, 1),
                                  # few input channels: the direct backward-data kernel (phases of the stride)
                                  (2, 4, 17, 19, 128, 3, 3, 1, 1), (3, 2, 20, 23, 16, 5, 5, 3, 2),
                                  (1, 3, 64, 300, 64, 7, 7, 2, 3),
                                  # ... its channel batches of 8: an odd count, a tail, one batch and a tail
                                  (2, 3, 13, 17, 24, 3, 3, 2, 1), (1, 2, 11, 9, 20, 4, 4, 2, 1),
                                  (2, 1, 9, 9, 12, 3, 3, 1, 1),
                                  # strided backward-data by phase of the stride (C > 4): stride 3 with a 5 x 5 filter,
                                  # stride 4 with 1 x 1 (15 of 16 phases have no tap), even filter, no padding + odd width,
                                  # a stride larger than the filter's reach
                                  (2, 8, 11, 13, 16, 5, 5, 3, 2), (1, 8, 9, 9, 8, 1, 1, 4, 0), (2, 16, 10, 10, 16, 2, 2, 2, 0),
                                  (1, 8, 9, 11, 8, 3, 3, 2, 0), (2, 64, 9, 9, 64, 2, 2, 3, 1),
                                  # ... enough 64 x 64 tiles for the large-tile kernels: the four phases' GEMMs as ONE
                                  # grouped launch (test_strided_bp_takes_the_grouped_launch pins the tile edge)
                                  (16, 128, 40, 40, 64, 3, 3, 2, 1),
                                  # strided forward over phase planes (64-channel blocks, more than one tap): odd sizes
                                  # without padding, a 5 x 3 filter with padding 2, an even filter, stride 4 and stride 3
                                  (2, 64, 11, 13, 64, 3, 3, 2, 0), (1, 128, 9, 10, 64, 5, 3, 2, 2), (2, 64, 12, 7, 64, 2, 2, 2, 1),
                                  (1, 64, 17, 19, 64, 5, 5, 4, 2), (2, 64, 16, 16, 128, 3, 3, 3, 1),
                                  # weight update on the large-tile kernels with OH*OW = 196 not a multiple of 32: both
                                  # operands padded to 224 (the separate-rounding variant pads in most cases above)
                                  (3, 256, 14, 14, 256, 3, 3, 1, 1)])
def test_conv_vs_oracle(hip, bname, variant, fused, case):
    """Batches with C or K < 64 included: every image is convolved (the reference collapses them)."""
    N, C, H, W, K, R, S, st, pd = case
    rng = np.random.default_rng(sum(case))
    OH, OW = (H + 2 * pd - R) // st + 1, (W + 2 * pd - S) // st + 1
    x = rng.standard_normal((N, C, H, W)).astype(np.float32)
    w = (rng.standard_normal((K, C, R, S)) * 0.3).astype(np.float32)
    b = rng.standard_normal(K).astype(np.float32)
    b[0] = -0.0
    go = rng.standard_normal((N, K, OH, OW)).astype(np.float32)
    y, gi, gw = _run_all(hip, x, w, b, go, st, pd, variant)
    assert_bits_equal(y, orc.conv2d_fp(x, w, b, st, pd, fused), "%s %s fp" % (bname, case))
    assert_bits_equal(gi, orc.conv2d_bp(go, w, (H, W), st, pd, fused), "%s %s bp" % (bname, case))
    assert_bits_equal(gw, orc.conv2d_wu(go, x, (R, S), st, pd, fused), "%s %s wu" % (bname, case))
    y0, _, _ = _run_all(hip, x, w, None, go, st, pd, variant)
    assert_bits_equal(y0, orc.conv2d_fp(x, w, None, st, pd, fused), "no bias")


@pytest.mark.parametrize("bname,variant,fused", VARIANTS)
@pytest.mark.parametrize("case", [(2, 3, 9, 9, 8, 3, 3, 2, 1), (1, 64, 8, 8, 64, 3, 3, 2, 1), (1, 64, 6, 6, 64, 3, 3, 1, 1)])
def test_conv_fp_signed_zero_chains(hip, bname, variant, fused, case):
    """Chains made of -0 products only: a -0 bias survives them where the reference has no channel padding (C a
    multiple of 64) and is +0 where it has (C < 64: also the case whose chains get a zero tail up to a whole k tile)."""
    N, C, H, W, K, R, S, st, pd = case
    x = np.zeros((N, C, H, W), np.float32)
    w = -np.ones((K, C, R, S), np.float32)
    b = np.zeros(K, np.float32)
    b[::2] = -0.0
    go = np.zeros((N, K, (H + 2 * pd - R) // st + 1, (W + 2 * pd - S) // st + 1), np.float32)
    y, _, _ = _run_all(hip, x, w, b, go, st, pd, variant)
    want = orc.conv2d_fp(x, w, b, st, pd, fused)
    assert_bits_equal(y, want, "%s %s" % (bname, case))
    assert (np.signbit(want).any()) == (C % 64 == 0)


def test_implicit_conv_small_workspace_chunks_and_fallback(hip):
    """The implicit path with a workspace that holds the padded planes of two images only (chunked), and with one
    too small for any padded image but big enough for the gathered form of one image... which is larger, so: the
    smallest workspace the gathered form accepts still serves the implicit path.  Same bits every way."""
    N, C, H, W, K, R, S, st, pd = 5, 64, 12, 12, 64, 3, 3, 1, 1
    rng = np.random.default_rng(6)
    x = rng.standard_normal((N, C, H, W)).astype(np.float32)
    w = (rng.standard_normal((K, C, R, S)) * 0.2).astype(np.float32)
    go = rng.standard_normal((N, K, H, W)).astype(np.float32)
    L = hip.lib()
    dims = (N, C, H, W, K, R, S, st, pd)
    want_fp = orc.conv2d_fp(x, w, None, st, pd, True)
    want_bp = orc.conv2d_bp(go, w, (H, W), st, pd, True)
    tx, tw, tg = _t(x), _t(w), _t(go)
    wf = K * C * R * S
    two_padded = (wf + 2 * C * (H + 2) * (W + 2) + 128 + R * (W + 2) + S + 4) * 4
    for nbytes in (L.fpemu_conv2d_workspace_bytes(0, *dims), L.fpemu_conv2d_workspace_bytes(0, 1, *dims[1:]),
                   two_padded):
        ws = torch.empty(nbytes // 4 + 4, device=DEV)
        y = torch.empty(N, K, H, W, device=DEV)
        gi = torch.empty(N, C, H, W, device=DEV)
        assert L.fpemu_conv2d_fp_f32(y.data_ptr(), tx.data_ptr(), tw.data_ptr(), None, *dims, 1, ws.data_ptr(),
                                     nbytes, None) == 0
        assert L.fpemu_conv2d_bp_f32(gi.data_ptr(), tg.data_ptr(), tw.data_ptr(), *dims, 1, ws.data_ptr(), nbytes,
                                     None) == 0
        torch.cuda.synchronize()
        assert_bits_equal(y.cpu().numpy(), want_fp, "implicit fp, workspace %d" % nbytes)
        assert_bits_equal(gi.cpu().numpy(), want_bp, "implicit bp, workspace %d" % nbytes)


def test_conv_small_workspace_chunks(hip):
    """A workspace that holds one image only: the batch is processed in chunks, same bits."""
    N, C, H, W, K, R, S, st, pd = 5, 8, 12, 12, 16, 3, 3, 1, 1
    rng = np.random.default_rng(5)
    x = rng.standard_normal((N, C, H, W)).astype(np.float32)
    w = rng.standard_normal((K, C, R, S)).astype(np.float32)
    L = hip.lib()
    dims = (N, C, H, W, K, R, S, st, pd)
    full = L.fpemu_conv2d_workspace_bytes(0, *dims)
    one = L.fpemu_conv2d_workspace_bytes(0, 1, *dims[1:])
    assert one < full
    ws = torch.empty(one // 4, device=DEV)
    tx, tw = _t(x), _t(w)
    y = torch.empty(N, K, H, W, device=DEV)
    assert L.fpemu_conv2d_fp_f32(y.data_ptr(), tx.data_ptr(), tw.data_ptr(), None, *dims, 1, ws.data_ptr(), one,
                                 None) == 0
    torch.cuda.synchronize()
    assert_bits_equal(y.cpu().numpy(), orc.conv2d_fp(x, w, None, st, pd, True), "chunked")
    assert L.fpemu_conv2d_fp_f32(y.data_ptr(), tx.data_ptr(), tw.data_ptr(), None, *dims, 1, ws.data_ptr(), 64,
                                 None) == -1
    assert L.fpemu_conv2d_fp_f32(y.data_ptr(), tx.data_ptr(), tw.data_ptr(), None, 1, 96, 4, 4, 8, 1, 1, 1, 0, 1,
                                 ws.data_ptr(), one, None) == -1   # C = 96: undefined in the reference


@pytest.mark.parametrize("bname,variant,fused", VARIANTS)
def test_conv_wu_chunked_padded_segments(hip, bname, variant, fused):
    """Weight update with a workspace for two images of five: the chain continues from chunk to chunk.  7 x 7 planes
    (49 products per image): the separate-rounding variant pads its operands to 64 per image (large-tile kernel), the
    MFMA variant here takes the small-tile kernel, unpadded -- same bits as the oracle either way, and as one chunk."""
    N, C, H, W, K, R, S, st, pd = 5, 64, 7, 7, 64, 3, 3, 1, 1
    rng = np.random.default_rng(12)
    x = rng.standard_normal((N, C, H, W)).astype(np.float32)
    go = rng.standard_normal((N, K, H, W)).astype(np.float32)
    L = hip.lib()
    dims = (N, C, H, W, K, R, S, st, pd)
    want = orc.conv2d_wu(go, x, (R, S), st, pd, fused)
    tx, tg = _t(x), _t(go)
    full = L.fpemu_conv2d_workspace_bytes(2, *dims)
    two = L.fpemu_conv2d_workspace_bytes(2, 2, *dims[1:])
    assert two < full
    for nbytes in (full, two):
        ws = torch.empty(nbytes // 4, device=DEV)
        gw = torch.empty(K, C, R, S, device=DEV)
        assert L.fpemu_conv2d_wu_f32(gw.data_ptr(), tg.data_ptr(), tx.data_ptr(), *dims, variant, ws.data_ptr(), nbytes,
                                     None) == 0, L.fpemu_last_error()
        torch.cuda.synchronize()
        assert_bits_equal(gw.cpu().numpy(), want, "%s workspace %d" % (bname, nbytes))


def test_simple_conv2d_module_api(hip):
    """mpemu.cmodel.simple.conv2d with autograd, the call torch.nn.functional.conv2d is patched to."""
    from mpemu.cmodel import simple

    rng = np.random.default_rng(8)
    x = rng.standard_normal((2, 8, 10, 10)).astype(np.float32)
    w = rng.standard_normal((16, 8, 3, 3)).astype(np.float32)
    b = rng.standard_normal(16).astype(np.float32)
    tx, tw, tb = _t(x).requires_grad_(True), _t(w).requires_grad_(True), _t(b)
    y = simple.conv2d(tx, tw, tb, stride=(2, 2), padding=(1, 1), dilation=(1, 1), groups=1)
    assert_bits_equal(y.detach().cpu().numpy(), orc.conv2d_fp(x, w, b, 2, 1, True), "conv2d")
    go = rng.standard_normal(tuple(y.shape)).astype(np.float32)
    y.backward(_t(go))
    assert_bits_equal(tx.grad.cpu().numpy(), orc.conv2d_bp(go, w, (10, 10), 2, 1, True), "grad input")
    assert_bits_equal(tw.grad.cpu().numpy(), orc.conv2d_wu(go, x, (3, 3), 2, 1, True), "grad weight")
    ref = torch.nn.functional.conv2d(_t(x), _t(w), tb, stride=2, padding=1)
    assert torch.allclose(y.detach(), ref, rtol=1e-4, atol=1e-4)
    # only the gradients autograd asks for are computed: images that need none (a first layer), a frozen weight
    ux, uw = _t(x), _t(w).requires_grad_(True)
    simple.conv2d(ux, uw, tb, stride=(2, 2), padding=(1, 1), dilation=(1, 1), groups=1).backward(_t(go))
    assert ux.grad is None
    assert_bits_equal(uw.grad.cpu().numpy(), orc.conv2d_wu(go, x, (3, 3), 2, 1, True), "grad weight, input needs none")
    vx, vw = _t(x).requires_grad_(True), _t(w)
    simple.conv2d(vx, vw, tb, stride=(2, 2), padding=(1, 1), dilation=(1, 1), groups=1).backward(_t(go))
    assert vw.grad is None
    assert_bits_equal(vx.grad.cpu().numpy(), orc.conv2d_bp(go, w, (10, 10), 2, 1, True), "grad input, frozen weight")
    for bad in (dict(dilation=(2, 2)), dict(padding=(1, 0)), dict(groups=2), dict(stride=(1, 2))):
        with pytest.raises(ValueError):
            simple.conv2d(tx, tw, tb, **bad)


def test_hw_patch_simple_routes_conv_and_matmul_through_cmodel(hip):
    """mpt_emu.quantize_model(..., hw_patch="simple"): inside a hooked module's forward F.conv2d and
    torch.matmul are the c-model ops (reference qutils.py:489-509), restored afterwards."""
    import torch.nn.functional as F

    from mpemu import mpt_emu
    from mpemu.cmodel import simple
    from mpemu.module_wrappers import Matmul

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.conv = torch.nn.Conv2d(64, 64, 3, padding=1)
            self.small = torch.nn.Conv2d(64, 8, 1)     # < 64 channels: left unpatched (e4m3_emu.py:99-104)
            self.mm = Matmul()

        def forward(self, x):
            y = self.conv(x)
            z = self.small(y)
            return self.mm(z.flatten(2), z.flatten(2).transpose(1, 2).contiguous())

    torch.manual_seed(3)
    net = Net().to(DEV).eval()
    seen = []
    real_conv, real_matmul = simple.conv2d, simple.matmul

    def spy_conv(input, weight, bias=None, *a, **k):
        out = real_conv(input, weight, bias, *a, **k)
        seen.append(("conv", input.detach().clone(), weight.detach().clone(), bias.detach().clone(), out.detach().clone()))
        return out

    simple.conv2d = spy_conv
    try:
        orig_conv, orig_matmul = F.conv2d, torch.matmul
        model, emu = mpt_emu.quantize_model(net, dtype="e4m3", hw_patch="simple", device="cuda")
        assert "small" in emu.emulator.list_unpatched and "conv" not in emu.emulator.list_unpatched
        x = torch.randn(2, 64, 6, 6, device=DEV)
        with torch.no_grad():
            out = model(x)
        assert F.conv2d is orig_conv and torch.matmul is orig_matmul and torch.Tensor.matmul is not simple.matmul
    finally:
        simple.conv2d = real_conv
    assert len(seen) == 1 and torch.isfinite(out).all()
    _, xi, wi, bi, yo = seen[0]
    want = orc.conv2d_fp(xi.cpu().numpy(), wi.cpu().numpy(), bi.cpu().numpy(), 1, 1, True)
    assert_bits_equal(yo.cpu().numpy(), want, "patched conv = c-model chain on the quantised operands")


# ---- the forward convolution's output maximum (fpemu_conv2d_fp_f32_amax) ----
@pytest.mark.parametrize("bname,variant,fused", VARIANTS)
@pytest.mark.parametrize("case", [(4, 64, 14, 14, 64, 3, 3, 1, 1),      # implicit GEMM, padded planes (fma variant)
                                  (3, 128, 9, 13, 64, 1, 1, 1, 0),      # implicit GEMM straight from the source planes
                                  (40, 64, 28, 28, 256, 3, 3, 1, 1),    # 128 x 128 tiles
                                  (2, 3, 32, 32, 64, 7, 7, 2, 3),       # gathered form, ragged rows (nstore)
                                  (1, 64, 28, 28, 192, 3, 3, 2, 1),     # gathered, strided
                                  (5, 1, 5, 5, 1, 5, 5, 1, 0)])         # one output per image
def test_conv_fp_amax_epilogue(hip, bname, variant, fused, case):
    """Same output bit for bit, and the word is bits(max |out|) over VALID outputs only (the implicit GEMM's garbage
    columns and the gathered form's zero-padded columns are computed but never stored, and never counted)."""
    N, C, H, W, K, R, S, st, pd = case
    rng = np.random.default_rng(sum(case) + 1)
    OH, OW = (H + 2 * pd - R) // st + 1, (W + 2 * pd - S) // st + 1
    x = _t(rng.standard_normal((N, C, H, W)).astype(np.float32))
    w = _t((rng.standard_normal((K, C, R, S)) * 0.3).astype(np.float32))
    b = _t(rng.standard_normal(K).astype(np.float32))
    for bias in (b, None):
        plain = torch.empty(N, K, OH, OW, device=DEV)
        withmax = torch.empty(N, K, OH, OW, device=DEV)
        word = torch.zeros(1, dtype=torch.int32, device=DEV)
        hip.conv2d_fp_(plain, x, w, bias, st, pd, variant)
        hip.conv2d_fp_(withmax, x, w, bias, st, pd, variant, amax=word)
        assert torch.equal(plain.view(torch.int32), withmax.view(torch.int32))
        assert int(word.item()) == int(np.float32(plain.abs().max().item()).view(np.uint32))


def test_hooked_conv_output_quantised_from_the_epilogue_maximum(hip, monkeypatch):
    """A hooked Conv2d with patch_ops / SIMPLE and a per-tensor output configuration (the hybrid training preset's
    oact): the c-model convolution leaves max |out| with the binding and the output hook's quantize_tensor is the cast
    alone.  Bits equal the run with the collection off, and the oracle's conv -> amax -> cast."""
    from mpemu import qutils
    from mpemu.cmodel import simple

    torch.manual_seed(5)
    conv = torch.nn.Conv2d(64, 64, 3, padding=1).to(DEV)
    x = torch.randn(2, 64, 12, 12, device=DEV)
    qc = qutils.ModuleQuantConfig(wt_qconfig=qutils.TensorQuantConfig("e4m3", "rne", "per-tensor"),
                                  iact_qconfig=qutils.TensorQuantConfig("e4m3", "rne", "per-tensor"),
                                  oact_qconfig=qutils.TensorQuantConfig("e4m3", "rne", "per-tensor"), patch_ops=True)
    qc.patch_impl = "SIMPLE"
    cfg = {"": qc}
    qutils.reset_quantization_setup(conv, cfg)
    handles = qutils.add_quantization_hooks(conv, cfg)
    w0 = conv.weight.data.clone()
    calls = []
    real = hip.cast_scaled_
    monkeypatch.setattr(hip, "cast_scaled_", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    try:
        with torch.no_grad():
            y_fast = conv(x).clone()
        assert calls == [1] and not hip.has_known_amax() and simple._collect is False
        conv.weight.data.copy_(w0)
        monkeypatch.setattr(simple, "collect_output_amax", lambda on: False)   # collection off: the ordinary path
        with torch.no_grad():
            y_plain = conv(x).clone()
        assert calls == [1]
    finally:
        for h in handles:
            h.remove()
    assert torch.equal(y_fast.view(torch.int32), y_plain.view(torch.int32))
    q = lambda a: orc.cast(a, "E4M3_RNE", scale=orc.scale_from_amax(orc.amax(a), 448.0))
    out = orc.conv2d_fp(q(x.cpu().numpy()), q(w0.cpu().numpy()), conv.bias.detach().cpu().numpy(), 1, 1, True)
    assert_bits_equal(y_fast.cpu().numpy(), q(out), "hooked conv output")


def test_strided_bp_takes_the_grouped_launch(hip):
    """The case (16, 128, 40, 40, 64, 3, 3, 2, 1) of test_conv_vs_oracle is there for the grouped launch of the phases'
    GEMMs (fpemu_gemm_chain_group_launch_): it needs the large-tile kernels, i.e. a tile edge of at least 64 for the
    128 x 448 (400 pixels per phase, padded) x 16-image problems, and whole tiles."""
    L = hip.lib()
    import ctypes
    L.fpemu_gemm_tile_edge_.argtypes = [ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int]
    L.fpemu_gemm_tile_edge_.restype = ctypes.c_int
    edge = L.fpemu_gemm_tile_edge_(128, 400, 16, 1)
    assert edge >= 64 and 128 % edge == 0


WU_IMPLICIT_CASES = [
    (3, 64, 8, 8, 64, 3, 3, 1, 1),      # 16 x 16 tiles, 64 pixels per image: every k tile ragged (masked loads)
    (2, 64, 16, 8, 64, 3, 3, 1, 1),     # 128 pixels per image: whole k tiles (unchecked loads)
    (2, 64, 12, 12, 64, 1, 1, 1, 0),    # 1 x 1, no padding: the input planes are the operand
    (2, 64, 14, 14, 64, 3, 3, 1, 0),    # no padding, 3 x 3: rows of 12 pixels in planes 14 wide
    (2, 8, 10, 16, 24, 3, 3, 1, 1),     # 8-channel block, 72 x 24 results: ragged tiles in both directions
    (2, 128, 8, 8, 128, 3, 3, 1, 1),    # 32 x 32 tiles
    (2, 128, 16, 16, 128, 3, 3, 1, 1),  # 32 x 32 tiles, whole k tiles
    (3, 64, 8, 12, 64, 5, 5, 1, 2),     # 5 x 5, padding 2
    (2, 192, 12, 8, 64, 3, 3, 1, 1),    # three channel blocks
    # strided: the same loader over the de-interleaved phase planes of the padded input (strides 2, 3, 4)
    (2, 64, 16, 16, 64, 3, 3, 2, 1), (2, 128, 16, 24, 128, 3, 3, 2, 1), (2, 64, 16, 22, 64, 3, 3, 3, 1),
    (1, 64, 17, 31, 64, 5, 5, 4, 2), (3, 64, 9, 16, 64, 2, 2, 2, 0),
]


@pytest.mark.parametrize("case", WU_IMPLICIT_CASES)
def test_conv_wu_implicit_operand(hip, case):
    """Weight update on the small-tile chain kernels: the loader waves read the (padded) input planes -- for strides 2-4
    their de-interleaved phase planes -- instead of a gathered matrix (fpemu_gemm_wu_implicit_launch_).  Same chains as the oracle's, bit for bit -- with an Inf in the
    input (a filler product of 0 x Inf would show as NaN), with chains of -0 products only, and chunk by chunk."""
    N, C, H, W, K, R, S, st, pd = case
    OH, OW = (H + 2 * pd - R) // st + 1, (W + 2 * pd - S) // st + 1
    L = hip.lib()
    rng = np.random.default_rng(sum(case) + 1)
    x = rng.standard_normal((N, C, H, W)).astype(np.float32)
    go = rng.standard_normal((N, K, OH, OW)).astype(np.float32)
    tg = _t(go)
    import ctypes

    L.fpemu_gemm_wu_implicit_ok_.restype = ctypes.c_int
    L.fpemu_gemm_wu_implicit_ok_.argtypes = [ctypes.c_void_p] + [ctypes.c_int64] * 5 + [ctypes.c_int]
    assert L.fpemu_gemm_wu_implicit_ok_(tg.data_ptr(), K, C * R * S, OH * OW, OW, K * OH * OW, 1) == 1, "case must qualify"
    want = orc.conv2d_wu(go, x, (R, S), st, pd, True)
    gw = torch.full((K, C, R, S), 7.0, device=DEV)
    hip.conv2d_wu_(gw, tg, _t(x), st, pd, 1)
    assert_bits_equal(gw.cpu().numpy(), want, "implicit wu %s" % (case,))
    # an Inf in the input, gout positive: every chain that meets it ends at +Inf, none at NaN
    xi, gp = x.copy(), np.abs(go) + 0.5
    xi[N - 1, C // 2, H // 2, W // 2] = np.inf
    xi = np.abs(xi)
    want = orc.conv2d_wu(gp, xi, (R, S), st, pd, True)
    assert np.isinf(want).any() and not np.isnan(want).any()
    hip.conv2d_wu_(gw, _t(gp), _t(xi), st, pd, 1)
    assert_bits_equal(gw.cpu().numpy(), want, "implicit wu, Inf %s" % (case,))
    # products of -0 only: chains from +0 stay +0
    xz, gn = np.zeros_like(x), -np.abs(go) - 1.0
    hip.conv2d_wu_(gw, _t(gn), _t(xz), st, pd, 1)
    assert_bits_equal(gw.cpu().numpy(), orc.conv2d_wu(gn, xz, (R, S), st, pd, True), "implicit wu, -0 products")
    # a workspace for one image: one launch per image, each continuing the chains of the one before
    dims = (N, C, H, W, K, R, S, st, pd)
    one = L.fpemu_conv2d_workspace_bytes(2, 1, *dims[1:])
    ws = torch.empty(one // 4, device=DEV)
    tx = _t(x)
    assert L.fpemu_conv2d_wu_f32(gw.data_ptr(), tg.data_ptr(), tx.data_ptr(), *dims, 1, ws.data_ptr(), one, None) == 0, \
        L.fpemu_last_error()
    torch.cuda.synchronize()
    assert_bits_equal(gw.cpu().numpy(), orc.conv2d_wu(go, x, (R, S), st, pd, True), "implicit wu, one image per chunk")


@pytest.mark.parametrize("case", [(5, 64, 12, 12, 64, 3, 3, 2, 1), (3, 64, 12, 12, 64, 2, 2, 2, 1), (3, 128, 9, 11, 64, 3, 3, 3, 0)])
def test_strided_conv_small_workspaces(hip, case):
    """Strided forward / weight update over phase planes with the full workspace, with one that holds a single image
    (chunked: the weight update's chains continue from launch to launch) and -- 2 x 2 filters, whose phase planes are
    larger than their gathered operand -- with one the phase planes do not fit in (the gathered form takes over)."""
    N, C, H, W, K, R, S, st, pd = case
    OH, OW = (H + 2 * pd - R) // st + 1, (W + 2 * pd - S) // st + 1
    rng = np.random.default_rng(sum(case) + 3)
    x = rng.standard_normal((N, C, H, W)).astype(np.float32)
    w = (rng.standard_normal((K, C, R, S)) * 0.2).astype(np.float32)
    b = rng.standard_normal(K).astype(np.float32)
    go = rng.standard_normal((N, K, OH, OW)).astype(np.float32)
    L = hip.lib()
    want_fp = orc.conv2d_fp(x, w, b, st, pd, True)
    want_wu = orc.conv2d_wu(go, x, (R, S), st, pd, True)
    tx, tw, tb, tg = _t(x), _t(w), _t(b), _t(go)
    for kind, full, one in ((0, L.fpemu_conv2d_workspace_bytes(0, *case), L.fpemu_conv2d_workspace_bytes(0, 1, *case[1:])),
                            (2, L.fpemu_conv2d_workspace_bytes(2, *case), L.fpemu_conv2d_workspace_bytes(2, 1, *case[1:]))):
        for nbytes in (full, one):
            ws = torch.empty(nbytes // 4 + 4, device=DEV)
            if kind == 0:
                y = torch.empty(N, K, OH, OW, device=DEV)
                assert L.fpemu_conv2d_fp_f32(y.data_ptr(), tx.data_ptr(), tw.data_ptr(), tb.data_ptr(), *case, 1, ws.data_ptr(),
                                             nbytes, None) == 0, L.fpemu_last_error()
                torch.cuda.synchronize()
                assert_bits_equal(y.cpu().numpy(), want_fp, "strided fp %s, workspace %d" % (case, nbytes))
            else:
                gw = torch.empty(K, C, R, S, device=DEV)
                assert L.fpemu_conv2d_wu_f32(gw.data_ptr(), tg.data_ptr(), tx.data_ptr(), *case, 1, ws.data_ptr(), nbytes,
                                             None) == 0, L.fpemu_last_error()
                torch.cuda.synchronize()
                assert_bits_equal(gw.cpu().numpy(), want_wu, "strided wu %s, workspace %d" % (case, nbytes))
