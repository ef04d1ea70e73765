"""fpemu_quantize_per_tensor_static -- the entry point that allocates nothing (include/fpemu_hip.h) --
against fpemu_quantize_per_tensor_ws, the entry point it is a variant of, through the C ABI.

Bit-exact bar (integer / bit work).  Two calls in a row on the shared scratch, the second with a smaller
input, across the size regimes of the per-tensor path (one workgroup; per-block maxima with 256 and with fewer
partials; the single-word path above 768 MiB is the same code as the _ws entry point's, reached with the same
word).  Collected last (test_zz_*): the round-end run uses pytest -x."""

import numpy as np
import pytest
import torch

import orc
from _util import assert_bits_equal, mixed_inputs

pytestmark = pytest.mark.gpu

E4M3, E5M2 = 1, 0
RNE, SR = 1, 2


def _finite(n, seed):
    x = mixed_inputs(n, seed=seed)
    x[~np.isfinite(x)] = 0.25
    return x


def _both(L, x, fmt, rm, flt_max, dtype_id=0, want_amax=True):
    """(static result, ws result, static amax word, ws amax word) for one input."""
    dev = torch.device("cuda", 0)
    t = torch.from_numpy(x).to(dev)
    a, b = torch.empty_like(t), torch.empty_like(t)
    ws = torch.full((257,), 0x5A5A5A5A, dtype=torch.int32, device=dev)  # need not be initialised
    word = torch.full((1,), 0x77777777, dtype=torch.int32, device=dev)
    s = torch.cuda.current_stream(dev).cuda_stream
    rc = L.fpemu_quantize_per_tensor_static(t.data_ptr(), a.data_ptr(), t.numel(), dtype_id, fmt, rm, flt_max, 1,
                                            word.data_ptr() if want_amax else None, 11, 5, s)
    assert rc == 0, L.fpemu_last_error()
    rc = L.fpemu_quantize_per_tensor_ws(t.data_ptr(), b.data_ptr(), t.numel(), dtype_id, fmt, rm, flt_max, 1,
                                        ws.data_ptr(), 257, 11, 5, s)
    assert rc == 0, L.fpemu_last_error()
    torch.cuda.synchronize()
    return a.cpu().numpy(), b.cpu().numpy(), int(word.item()), int(ws[0].item())


@pytest.mark.parametrize("fmt,rm,flt_max", [(E4M3, RNE, 448.0), (E5M2, SR, 57344.0)])
def test_static_equals_ws_two_calls_second_smaller(fmt, rm, flt_max):
    from mpemu.pytquant.hip import _binding as hip

    L = hip.lib()
    sizes = [(1 << 22) + 777, 5000,            # per-block maxima, then one workgroup
             (1 << 24) + 3, (1 << 16) + 1,     # 256 partials, then 64 partials: stale maxima must not be read
             8192, 17]
    for k, n in enumerate(sizes):
        x = _finite(n, seed=100 + k)
        if k % 2 == 0:
            x[n // 3] = 3.0e6  # the bigger call leaves a large maximum behind in the scratch
        got, want, w_static, w_ws = _both(L, x, fmt, rm, flt_max)
        assert_bits_equal(got, want, "static vs ws, n=%d" % n)
        assert w_static == w_ws, (n, hex(w_static), hex(w_ws))
        assert w_ws == int(np.abs(x).max().view(np.uint32)), n
        # and against the oracle itself (seed 11, call offset 5 as passed above), not only the sibling entry point
        mode = "E4M3_RNE" if (fmt, rm) == (E4M3, RNE) else "E5M2_STOCHASTIC"
        scale = orc.scale_from_amax(orc.amax(x), flt_max)
        assert_bits_equal(got, orc.cast(x, mode, scale=scale, seed=11, offset=5), "static vs oracle, n=%d" % n)


def test_static_without_amax_word_and_16bit_storage():
    from mpemu.pytquant.hip import _binding as hip

    L = hip.lib()
    for n in ((1 << 20) + 8, 4096):
        x = np.clip(_finite(n, seed=7), -6.0e4, 6.0e4)
        dev = torch.device("cuda", 0)
        t = torch.from_numpy(x).to(dev).to(torch.bfloat16)
        a, b = torch.empty_like(t), torch.empty_like(t)
        ws = torch.empty(257, dtype=torch.int32, device=dev)
        s = torch.cuda.current_stream(dev).cuda_stream
        assert L.fpemu_quantize_per_tensor_static(t.data_ptr(), a.data_ptr(), n, 2, E4M3, RNE, 448.0, 1, None, 0, 0,
                                                  s) == 0
        assert L.fpemu_quantize_per_tensor_ws(t.data_ptr(), b.data_ptr(), n, 2, E4M3, RNE, 448.0, 1, ws.data_ptr(),
                                              257, 0, 0, s) == 0
        torch.cuda.synchronize()
        assert torch.equal(a.view(torch.int16), b.view(torch.int16)), n


def test_static_rejects_oversize_and_allocates_nothing():
    from mpemu.pytquant.hip import _binding as hip

    L = hip.lib()
    dev = torch.device("cuda", 0)
    t = torch.ones(1024, device=dev)
    s = torch.cuda.current_stream(dev).cuda_stream
    rc = L.fpemu_quantize_per_tensor_static(t.data_ptr(), t.data_ptr(), (1 << 36) + 1, 0, E4M3, RNE, 448.0, 1, None,
                                            0, 0, s)
    assert rc == -1 and b"FPEMU_STATIC_MAX_ELEMS" in L.fpemu_last_error()
    # device memory in use (driver view) does not move across calls: nothing is allocated
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info(dev)
    x = torch.randn(1 << 22, device=dev)
    y = torch.empty_like(x)
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info(dev)
    for _ in range(4):
        assert L.fpemu_quantize_per_tensor_static(x.data_ptr(), y.data_ptr(), x.numel(), 0, E4M3, RNE, 448.0, 1, None,
                                                  0, 0, s) == 0
    torch.cuda.synchronize()
    free2, _ = torch.cuda.mem_get_info(dev)
    assert free2 == free1, (free0, free1, free2)


def test_static_under_graph_capture():
    """The callers it is for: a capture that must not allocate."""
    from mpemu.pytquant.hip import _binding as hip

    L = hip.lib()
    dev = torch.device("cuda", 0)
    x = torch.from_numpy(_finite(1 << 20, seed=3)).to(dev)
    y = torch.zeros_like(x)
    want = torch.empty_like(x)
    ws = torch.empty(257, dtype=torch.int32, device=dev)
    s0 = torch.cuda.current_stream(dev).cuda_stream
    # first call outside the capture: the per-device symbol lookup happens here
    assert L.fpemu_quantize_per_tensor_static(x.data_ptr(), y.data_ptr(), x.numel(), 0, E4M3, RNE, 448.0, 1, None, 0, 0,
                                              s0) == 0
    assert L.fpemu_quantize_per_tensor_ws(x.data_ptr(), want.data_ptr(), x.numel(), 0, E4M3, RNE, 448.0, 1,
                                          ws.data_ptr(), 257, 0, 0, s0) == 0
    torch.cuda.synchronize()
    y.zero_()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        s = torch.cuda.current_stream(dev).cuda_stream
        assert L.fpemu_quantize_per_tensor_static(x.data_ptr(), y.data_ptr(), x.numel(), 0, E4M3, RNE, 448.0, 1, None,
                                                  0, 0, s) == 0
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(y.view(torch.int32), want.view(torch.int32))
