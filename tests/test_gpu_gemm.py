"""cmodel emulated GEMM on the GPU vs the reference's simple_gemm_dev (both of its builds, see
oracle/build_ref.py) and vs the oracle's k-ascending chains.  Bit-exact for both kernel variants:
GEMM_FMA (MFMA fp32, an fmaf chain) against the reference's FMA build, GEMM_SEPARATE (VALU, product
and sum rounded separately) against its no-FMA build."""
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


@pytest.mark.parametrize("bname,variant,fused", VARIANTS)
def test_gemm_vs_reference_fixture(hip, bname, variant, fused):
    g = npz("gemm.npz")
    for key in g["names"]:
        M, N, K, at, bt = map(int, str(key).split("_"))
        A = g["A/" + str(key)].view(np.float32)
        B = g["B/" + str(key)].view(np.float32)
        for alpha in (1.0, 1.5):
            C = torch.zeros(M, N, device=DEV)
            hip.gemm_(C, _t(A), _t(B), alpha, bool(at), bool(bt), variant)
            assert_bits_equal(C.cpu().numpy(), g["C/%s/%s/%g" % (bname, key, alpha)], "%s %s a=%g" % (bname, key, alpha))
        C = _t(np.arange(M * N, dtype=np.float32).reshape(M, N) * 0.25 - 3.0)
        hip.gemm_(C, _t(A), _t(B), 0.75, bool(at), bool(bt), variant)
        assert_bits_equal(C.cpu().numpy(), g["Cacc/%s/%s" % (bname, key)], "%s %s accumulate" % (bname, key))


@pytest.mark.parametrize("bname,variant,fused", VARIANTS)
@pytest.mark.parametrize("shape", [(1, 1, 1), (7, 5, 3), (128, 128, 16), (129, 127, 17), (256, 384, 512),
                                   (300, 200, 1000), (64, 1024, 33), (1000, 8, 2048),
                                   # few outputs, long k: the 16x16x4 chain kernel, odd / even tile counts,
                                   # whole tiles and ragged ones (steady loop + every tail form)
                                   (96, 160, 1664), (50, 70, 1500), (32, 32, 640), (33, 31, 129),
                                   # 128 .. 320 tiles of 32 x 32: the four-chain form of that kernel (fewer: one chain per workgroup)
                                   (384, 384, 256), (380, 372, 200), (16, 16, 4096), (48, 20, 130),
                                   (2950, 2944, 70), (2944, 2944, 64)])   # >= 512 tiles: the 128 x 128 kernels
@pytest.mark.parametrize("at,bt", [(False, False), (True, False), (False, True), (True, True)])
def test_gemm_vs_oracle(hip, bname, variant, fused, shape, at, bt):
    M, N, K = shape
    rng = np.random.default_rng(M * 7 + N * 3 + K)
    A = (rng.standard_normal((K, M) if at else (M, K)) * np.exp(rng.standard_normal((K, M) if at else (M, K)))).astype(np.float32)
    B = rng.standard_normal((N, K) if bt else (K, N)).astype(np.float32)
    A.flat[::97] = 0.0
    if A.size > 10:
        A.flat[5] = 1e-41       # denormal operand
        B.flat[3] = -0.0
    C = torch.zeros(M, N, device=DEV)
    hip.gemm_(C, _t(A), _t(B), 1.0, at, bt, variant)
    assert_bits_equal(C.cpu().numpy(), orc.gemm(A, B, 1.0, at, bt, fused=fused), "%s %s %s %s" % (bname, shape, at, bt))


def test_gemm_k0_and_errors(hip):
    C = torch.full((4, 4), 2.0, device=DEV)
    hip.gemm_(C, torch.zeros(4, 0, device=DEV), torch.zeros(0, 4, device=DEV))
    assert torch.equal(C, torch.full((4, 4), 2.0, device=DEV))   # C + (alpha*0 + 0*C)
    with pytest.raises(ValueError):
        hip.gemm_(C, torch.zeros(4, 3, device=DEV), torch.zeros(5, 4, device=DEV))
    with pytest.raises(TypeError):
        hip.gemm_(C, torch.zeros(4, 3, device=DEV).half(), torch.zeros(3, 4, device=DEV))
    with pytest.raises(ValueError):
        hip.gemm_(C, torch.zeros(4, 3, device=DEV), torch.zeros(3, 4, device=DEV), variant=7)


@pytest.mark.parametrize("bname,variant,fused", VARIANTS)
def test_gemm_2048_sampled(hip, bname, variant, fused):
    """Large case: full oracle compare on a sample of rows (the oracle runs ~4 GFLOP/s)."""
    M = N = K = 2048
    g = torch.Generator(device=DEV).manual_seed(5)
    A = torch.randn(M, K, device=DEV, generator=g)
    B = torch.randn(K, N, device=DEV, generator=g)
    C = torch.zeros(M, N, device=DEV)
    hip.gemm_(C, A, B, 1.0, False, False, variant)
    rows = [0, 1, 127, 128, 1000, 2047]
    want = orc.gemm(A[rows].cpu().numpy(), B.cpu().numpy(), 1.0, fused=fused)
    assert_bits_equal(C[rows].cpu().numpy(), want, bname + " 2048 rows")
    ref = A.double() @ B.double()
    assert ((C.double() - ref).abs().max() / ref.abs().max()).item() < 1e-5


@pytest.mark.parametrize("bname,variant,fused", VARIANTS)
@pytest.mark.parametrize("at,bt", [(False, False), (True, True)])
def test_gemm_many_ragged_tiles_sampled(hip, bname, variant, fused, at, bt):
    """More than 1024 tiles of 128 x 128 with ragged edges in M, N and k: the large-tile kernels' bounds-checked path
    (smaller problems take the 64 x 64 / 32 x 32 / 16 x 16 kernels).  Oracle on a sample of rows."""
    M, N, K = 4100, 4090, 37
    g = torch.Generator(device=DEV).manual_seed(11)
    A = torch.randn((K, M) if at else (M, K), device=DEV, generator=g)
    B = torch.randn((N, K) if bt else (K, N), device=DEV, generator=g)
    C = torch.zeros(M, N, device=DEV)
    hip.gemm_(C, A, B, 1.0, at, bt, variant)
    rows = [0, 127, 128, 2049, 4095, 4096, 4099]
    An = (A.t() if at else A)[rows].contiguous().cpu().numpy()
    Bn = (B.t() if bt else B).contiguous().cpu().numpy()
    assert_bits_equal(C[rows].cpu().numpy(), orc.gemm(An, Bn, 1.0, fused=fused), bname + " ragged rows")
    ref = (A.t() if at else A).double() @ (B.t() if bt else B).double()
    assert ((C.double() - ref).abs().max() / ref.abs().max()).item() < 1e-5


def test_simple_module_api(hip):
    """mpemu.cmodel.simple.{mm, matmul, addmm} keep the reference's call surface (simple.py:30-86)."""
    from mpemu.cmodel import simple

    rng = np.random.default_rng(1)
    a = rng.standard_normal((70, 90)).astype(np.float32)
    b = rng.standard_normal((90, 50)).astype(np.float32)
    bias = rng.standard_normal((50,)).astype(np.float32)
    ta, tb = _t(a).requires_grad_(True), _t(b).requires_grad_(True)
    assert simple.get_accumulation() == "fma"
    y = simple.mm(ta, tb)
    assert_bits_equal(y.detach().cpu().numpy(), orc.gemm(a, b, fused=True), "mm")
    # transposed views are detected and passed as flags, not copied (simple.py:21-27)
    y2 = simple.matmul(_t(a.T.copy()).t(), tb)
    assert_bits_equal(y2.detach().cpu().numpy(), orc.gemm(a.T.copy(), b, a_trans=True, fused=True), "matmul T")
    y3 = simple.addmm(_t(bias), ta, tb, beta=1.0, alpha=1.0)
    want = orc.gemm(a, b, fused=True) + bias
    assert_bits_equal(y3.detach().cpu().numpy(), want, "addmm")
    yb = simple.matmul(torch.stack([_t(a), _t(a * 2)]), tb)
    assert yb.shape == (2, 70, 50)
    assert_bits_equal(yb[1].detach().cpu().numpy(), orc.gemm(a * 2, b, fused=True), "batched matmul")
    # backward: grad_a = g @ b^T, grad_b = a^T @ g through the same cmodel GEMM
    gout = rng.standard_normal((70, 50)).astype(np.float32)
    y.backward(_t(gout))
    assert_bits_equal(ta.grad.cpu().numpy(), orc.gemm(gout, b, b_trans=True, fused=True), "grad a")
    assert_bits_equal(tb.grad.cpu().numpy(), orc.gemm(a, gout, a_trans=True, fused=True), "grad b")
    # only the gradients autograd asks for are computed (a frozen weight; activations that need none)
    ua, ub = _t(a).requires_grad_(True), _t(b)
    simple.mm(ua, ub).backward(_t(gout))
    assert ub.grad is None
    assert_bits_equal(ua.grad.cpu().numpy(), orc.gemm(gout, b, b_trans=True, fused=True), "grad a, frozen b")
    va, vb = _t(a), _t(b).requires_grad_(True)
    simple.addmm(_t(bias), va, vb, beta=1.0, alpha=1.0).backward(_t(gout))
    assert va.grad is None
    assert_bits_equal(vb.grad.cpu().numpy(), orc.gemm(a, gout, a_trans=True, fused=True), "grad b, a needs none")
    simple.set_accumulation("separate")
    try:
        assert_bits_equal(simple.mm(_t(a), _t(b)).cpu().numpy(), orc.gemm(a, b, fused=False), "separate")
    finally:
        simple.set_accumulation("fma")
    with pytest.warns(UserWarning):
        simple.mm(_t(a).double(), _t(b).double())


@pytest.mark.parametrize("variant,fused", [(1, True), (0, False)])
def test_gemm_chain_k_segments_and_batch(hip, variant, fused):
    """fpemu_gemm_chain_f32: accumulators start at C; k segments chain several (A, B) pairs into one chain;
    batch = independent problems.  Oracle: the same chain as one long-K GEMM started from C."""
    import ctypes

    rng = np.random.default_rng(4)
    nb, nseg, M, N, K = 3, 4, 70, 90, 50
    A = rng.standard_normal((nb, nseg, M, K)).astype(np.float32)
    B = rng.standard_normal((nb, nseg, K, N)).astype(np.float32)
    C0 = rng.standard_normal((nb, M, N)).astype(np.float32)
    tA, tB, tC = _t(A), _t(B), _t(C0)
    L = hip.lib()
    rc = L.fpemu_gemm_chain_f32(tC.data_ptr(), tA.data_ptr(), tB.data_ptr(), M, N, K, K, N, N, 0, 0, variant,
                                nb, nseg * M * K, nseg * K * N, M * N, nseg, M * K, K * N, None)
    assert rc == 0, L.fpemu_last_error()
    torch.cuda.synchronize()
    for b in range(nb):
        # chain from C0: out = C0 then fmaf / mul+add over the concatenated k -- the oracle's GEMM chain with
        # alpha = 1 starts at +0, so restate the chain directly
        want = C0[b].copy()
        Acat = np.concatenate([A[b, sgm] for sgm in range(nseg)], axis=1)
        Bcat = np.concatenate([B[b, sgm] for sgm in range(nseg)], axis=0)
        for k in range(Acat.shape[1]):
            a, bb = Acat[:, k:k + 1], Bcat[k:k + 1, :]
            if fused:
                want = (a.astype(np.float64) * bb.astype(np.float64) + want.astype(np.float64)).astype(np.float32)
            else:
                want = want + (a * bb)
        assert_bits_equal(tC[b].cpu().numpy(), want, "chain batch %d" % b)
    assert L.fpemu_gemm_chain_f32(tC.data_ptr(), tA.data_ptr(), tB.data_ptr(), M, N, K, K, N, N, 0, 0, variant,
                                  1, 0, 0, 0, 0, 0, 0, None) == -1   # k_segments < 1


@pytest.mark.parametrize("variant", [1, 0])
@pytest.mark.parametrize("M,N,K", [(16, 16, 64), (64, 16, 576), (256, 256, 64), (512, 512, 64), (1024, 1024, 64),
                                   (2048, 2048, 48), (48, 20, 130), (61, 99, 7)])
def test_chain_signed_zeros(hip, variant, M, N, K):
    """A chain sitting at -0 stays there through products of -0 and leaves it at the first +0 product, in every kernel
    (tile sizes by problem size) and with a ragged last k tile: the tile's filler products are -0, not +0."""
    L = hip.lib()
    rng = np.random.default_rng(M + N + K)
    A = -np.abs(rng.standard_normal((M, K))).astype(np.float32)      # negative
    B = np.zeros((K, N), np.float32)                                 # +0: every product -0
    first_pos = rng.integers(0, K + 1, size=N)                       # column j: from row first_pos[j] on, B is -0
    first_pos[::4] = K                                               # (every fourth column: never)
    for j in range(N):
        B[first_pos[j]:, j] = -0.0                                   # ... and those products +0
    C0 = np.full((M, N), -0.0, np.float32)
    C0[::3, ::5] = 0.0
    tA, tB, tC = _t(A), _t(B), _t(C0)
    rc = L.fpemu_gemm_chain_f32(tC.data_ptr(), tA.data_ptr(), tB.data_ptr(), M, N, K, K, N, N, 0, 0, variant, 1, 0, 0, 0,
                                1, 0, 0, None)
    assert rc == 0, L.fpemu_last_error()
    got = tC.cpu().numpy()
    want_neg = np.signbit(C0) & (first_pos == K)[None, :]             # -0 start and no +0 product in the column
    assert np.all(got == 0.0)
    assert np.array_equal(np.signbit(got), want_neg), "%d sign bits differ" % int((np.signbit(got) != want_neg).sum())
    assert want_neg.any() and not want_neg.all()


@pytest.mark.parametrize("M,N,K,nseg,nb", [
    (64, 96, 128, 3, 2),      # 16 x 16 tiles, whole tiles of aligned operands (unchecked loads)
    (60, 100, 132, 3, 2),     # 16 x 16 tiles, aligned operands with ragged edges in M, N and k (masked loads)
    (61, 99, 130, 2, 1),      # 16 x 16 tiles, unaligned (checked scalar loads)
    (256, 320, 128, 2, 2),    # 32 x 32 tiles (160 of them), whole
    (252, 316, 136, 2, 2),    # 32 x 32 tiles, ragged, aligned
    (250, 318, 70, 3, 2),     # 32 x 32 tiles, unaligned
    (16, 16, 3136, 8, 1),     # one chain, many ragged segments (the conv weight update's k = OH * OW per image)
])
@pytest.mark.parametrize("at,bt", [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_chain_kernel_forms_vs_oracle(hip, M, N, K, nseg, nb, at, bt):
    """The wave-specialised small-tile kernels behind fpemu_gemm_chain_f32, every load mode and both tile sizes, with k
    segments and batch: chain from C = 0 equals the oracle's GEMM over the concatenated k."""
    rng = np.random.default_rng(M * 31 + N * 7 + K + nseg)
    A = rng.standard_normal((nb, nseg, K, M) if at else (nb, nseg, M, K)).astype(np.float32)
    B = rng.standard_normal((nb, nseg, N, K) if bt else (nb, nseg, K, N)).astype(np.float32)
    A.flat[::53] = 0.0
    tA, tB = _t(A), _t(B)
    tC = torch.zeros(nb, M, N, device=DEV)
    L = hip.lib()
    lda, ldb = (M if at else K), (K if bt else N)
    rc = L.fpemu_gemm_chain_f32(tC.data_ptr(), tA.data_ptr(), tB.data_ptr(), M, N, K, lda, ldb, N, at, bt, 1,
                                nb, nseg * M * K, nseg * K * N, M * N, nseg, M * K, K * N, None)
    assert rc == 0, L.fpemu_last_error()
    got = tC.cpu().numpy()
    for b in range(nb):
        Acat = np.concatenate([A[b, s].T if at else A[b, s] for s in range(nseg)], axis=1)   # [M, nseg * K]
        Bcat = np.concatenate([B[b, s].T if bt else B[b, s] for s in range(nseg)], axis=0)   # [nseg * K, N]
        assert_bits_equal(got[b], orc.gemm(Acat, Bcat, 1.0, False, False, fused=True), "chain %s batch %d" % ((M, N, K, nseg, at, bt), b))


# ---- the epilogue's output maximum (fpemu_gemm_f32_amax; reference qutils.py:586-609 quantises the product next) ----
AMAX_SHAPES = [(1, 1, 1), (129, 127, 17), (256, 384, 512), (1000, 8, 2048), (96, 160, 1664), (33, 31, 129),
               (384, 384, 256), (16, 16, 4096), (2950, 2944, 70), (2944, 2944, 64), (2048, 2048, 64)]


@pytest.mark.parametrize("bname,variant,fused", VARIANTS)
@pytest.mark.parametrize("shape", AMAX_SHAPES)
def test_gemm_amax_epilogue(hip, bname, variant, fused, shape):
    """Every kernel behind fpemu_gemm_f32 (tile sizes by problem size, whole and ragged tiles): the _amax form writes the
    same C bit for bit and its word is bits(max |C|) of the FINAL C (alpha and the accumulate included); the word
    accumulates over calls; a NaN wins."""
    M, N, K = shape
    rng = np.random.default_rng(M + 3 * N + 7 * K)
    A = _t(rng.standard_normal((M, K)).astype(np.float32))
    B = _t(rng.standard_normal((K, N)).astype(np.float32))
    C0 = _t((rng.standard_normal((M, N)) * 4).astype(np.float32))
    plain, withmax = C0.clone(), C0.clone()
    word = torch.zeros(1, dtype=torch.int32, device=DEV)
    hip.gemm_(plain, A, B, 0.75, False, False, variant)
    hip.gemm_(withmax, A, B, 0.75, False, False, variant, amax=word)
    assert torch.equal(plain.view(torch.int32), withmax.view(torch.int32)), "the matrix changed"
    want = int(np.float32(plain.abs().max().item()).view(np.uint32))
    assert int(word.item()) == want, (hex(int(word.item())), hex(want))
    # a second product into the same word: the maximum of both
    small = torch.zeros(M, N, device=DEV)
    hip.gemm_(small, A * 1e-3, B, 1.0, False, False, variant, amax=word)
    assert int(word.item()) == want
    big = torch.zeros(M, N, device=DEV)
    hip.gemm_(big, A * 64.0, B, 1.0, False, False, variant, amax=word)
    assert int(word.item()) == int(np.float32(max(big.abs().max().item(), plain.abs().max().item())).view(np.uint32))
    # NaN: above every number, as torch.max
    A2 = A.clone()
    A2[M // 2, K // 2] = float("nan")
    word.zero_()
    hip.gemm_(torch.zeros(M, N, device=DEV), A2, B, 1.0, False, False, variant, amax=word)
    assert (int(word.item()) & 0x7FFFFFFF) > 0x7F800000


def test_gemm_amax_errors(hip):
    C = torch.zeros(4, 4, device=DEV)
    A = torch.zeros(4, 3, device=DEV)
    B = torch.zeros(3, 4, device=DEV)
    with pytest.raises(ValueError):
        hip.gemm_(C, A, B, amax=torch.zeros(1, device=DEV))                     # float word
    with pytest.raises(ValueError):
        hip.gemm_(C, A, B, amax=torch.zeros(1, dtype=torch.int32))              # host word
    L = hip.lib()
    assert L.fpemu_gemm_f32_amax(C.data_ptr(), A.data_ptr(), B.data_ptr(), 4, 4, 3, 3, 4, 4, 1.0, 0, 0, 1, None, None) == -1


@pytest.mark.parametrize("acc", ["fma", "separate"])
def test_simple_matmul_leaves_its_maximum_for_quantize_tensor(hip, acc, monkeypatch):
    """mpemu.cmodel.simple with collect_output_amax(True): quantize_tensor of the product (per-tensor max) takes the
    epilogue's word -- one cast, no amax pass -- and gives the bits of the ordinary path and of the oracle."""
    from mpemu import qutils
    from mpemu.cmodel import simple

    rng = np.random.default_rng(21)
    a = (rng.standard_normal((300, 260)) * 2).astype(np.float32)
    b = rng.standard_normal((260, 200)).astype(np.float32)
    cfg = qutils.TensorQuantConfig("e4m3", "rne", "per-tensor")
    simple.set_accumulation(acc)
    prev = simple.collect_output_amax(True)
    calls = []
    real = hip.cast_scaled_
    monkeypatch.setattr(hip, "cast_scaled_", lambda *a_, **k: (calls.append(1), real(*a_, **k))[1])
    try:
        y = simple.mm(_t(a), _t(b))
        q = qutils.quantize_tensor(y, cfg, None, inplace=False)
        assert calls == [1], "the known maximum was not used"
        prod = orc.gemm(a, b, fused=acc == "fma")
        want = orc.cast(prod, "E4M3_RNE", scale=orc.scale_from_amax(orc.amax(prod), 448.0))
        assert_bits_equal(q.cpu().numpy(), want, "quantised product")
        # a tensor that is NOT the one the op produced never takes the word
        y2 = simple.mm(_t(a), _t(b))
        other = y2.clone()
        qutils.quantize_tensor(other, cfg, None, inplace=False)
        assert calls == [1] and not hip.has_known_amax()
        # ... nor one changed in place since
        y3 = simple.mm(_t(a), _t(b))
        y3.mul_(2.0)
        q3 = qutils.quantize_tensor(y3, cfg, None, inplace=False)
        assert calls == [1]
        assert_bits_equal(q3.cpu().numpy(), orc.cast(prod * 2, "E4M3_RNE", scale=orc.scale_from_amax(orc.amax(prod * 2), 448.0)),
                          "changed in place")
        # a stack of matrices times a matrix: one word for all slices, in place this time
        yb = simple.matmul(torch.stack([_t(a), _t(a * 3)]), _t(b))
        qutils.quantize_tensor(yb, cfg, None, inplace=True)
        assert calls == [1, 1]
        pb = np.stack([prod, orc.gemm(a * 3, b, fused=acc == "fma")])
        assert_bits_equal(yb.cpu().numpy(), orc.cast(pb, "E4M3_RNE", scale=orc.scale_from_amax(orc.amax(pb), 448.0)), "stack")
    finally:
        simple.collect_output_amax(prev)
        simple.set_accumulation("fma")
        hip.forget_amax()


@pytest.mark.parametrize("bname,variant,fused", VARIANTS)
@pytest.mark.parametrize("shape", [(4100, 4092, 64), (1100, 1028, 64), (4096, 4092, 32), (4100, 4096, 96)])
@pytest.mark.parametrize("at,bt", [(False, False), (True, True), (False, True)])
def test_gemm_ragged_edges_interior_and_strips(hip, bname, variant, fused, shape, at, bt):
    """Ragged M and / or N with aligned operands and whole k tiles: the launch is cut into the interior of whole tiles
    (unchecked loads) plus a right and a bottom strip.  Whole matrix against the oracle, accumulate and alpha included,
    and the epilogue maximum over all three launches."""
    M, N, K = shape
    rng = np.random.default_rng(M + N + K + at + 2 * bt)
    A = rng.standard_normal((K, M) if at else (M, K)).astype(np.float32)
    B = rng.standard_normal((N, K) if bt else (K, N)).astype(np.float32)
    C0 = rng.standard_normal((M, N)).astype(np.float32)
    C = _t(C0)
    word = torch.zeros(1, dtype=torch.int32, device=DEV)
    hip.gemm_(C, _t(A), _t(B), 0.5, at, bt, variant, amax=word)
    prod = orc.gemm(A, B, 1.0, at, bt, fused=fused)
    want = C0 + (np.float32(0.5) * prod + np.float32(0.0) * C0)   # impl.cpp:226-231, element by element
    got = C.cpu().numpy()
    assert_bits_equal(got, want, "%s %s at=%d bt=%d" % (bname, shape, at, bt))
    assert int(word.item()) == int(np.abs(want).max().view(np.uint32))
