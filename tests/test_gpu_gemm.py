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
                                   (300, 200, 1000), (64, 1024, 33), (1000, 8, 2048)])
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
    simple.set_accumulation("separate")
    try:
        assert_bits_equal(simple.mm(_t(a), _t(b)).cpu().numpy(), orc.gemm(a, b, fused=False), "separate")
    finally:
        simple.set_accumulation("fma")
    with pytest.warns(UserWarning):
        simple.mm(_t(a).double(), _t(b).double())
