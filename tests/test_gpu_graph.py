"""The entry points only enqueue work on the caller's stream (no allocation, no sync), so a sequence of
them can be captured into a HIP graph and replayed on new data: launch-bound per-layer casts become one
graph launch."""
import numpy as np
import pytest
import torch

import orc
from _util import assert_bits_equal, mixed_inputs

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_cast_quantize_gemm_under_graph_capture():
    from mpemu.pytquant.hip import _binding as hip

    hip.lib()
    n = 70001
    x = torch.zeros(n, device=DEV)
    y = torch.empty_like(x)
    z = torch.empty_like(x)
    ws = torch.zeros(1, dtype=torch.int32, device=DEV)
    a = torch.zeros(96, 80, device=DEV)
    b = torch.zeros(80, 72, device=DEV)
    c = torch.zeros(96, 72, device=DEV)
    f4, r4 = hip.parse_mode("E4M3_RNE")
    f5, r5 = hip.parse_mode("E5M2_STOCHASTIC")

    def work():
        hip.cast_(x, y, f5, r5, scale=1.0, seed=3, offset=7)
        hip.quantize_per_tensor_(x, z, f4, r4, 448.0, amax_ws=ws)
        c.zero_()
        hip.gemm_(c, a, b, 1.0, False, False, hip.GEMM_FMA)

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        work()                       # warm-up outside capture
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        work()
    for seed in (1, 2):
        hx = mixed_inputs(n, seed=seed)
        hx[np.isnan(hx)] = 0.25
        rng = np.random.default_rng(seed)
        ha = rng.standard_normal((96, 80)).astype(np.float32)
        hb = rng.standard_normal((80, 72)).astype(np.float32)
        x.copy_(torch.from_numpy(hx))
        a.copy_(torch.from_numpy(ha))
        b.copy_(torch.from_numpy(hb))
        g.replay()
        torch.cuda.synchronize()
        assert_bits_equal(y.cpu().numpy(), orc.cast(hx, "E5M2_STOCHASTIC", seed=3, offset=7), "graph cast")
        sc = orc.scale_from_amax(orc.amax(hx), 448.0)
        assert_bits_equal(z.cpu().numpy(), orc.cast(hx, "E4M3_RNE", scale=sc), "graph per-tensor quantise",
                          nan_payload_ok=True)
        assert_bits_equal(c.cpu().numpy(), orc.gemm(ha, hb, fused=True), "graph gemm")


def test_profiler_ranges_when_tracing_is_on():
    """MPEMU_TRACE / set_tracing: native calls carry torch.profiler ranges (the reference marks its c-model
    ops with RECORD_FUNCTION)."""
    from mpemu.pytquant.hip import _binding as hip

    x = torch.randn(4096, device=DEV)
    a = torch.randn(64, 64, device=DEV)
    c = torch.zeros(64, 64, device=DEV)
    hip.set_tracing(True)
    try:
        with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU]) as prof:
            hip.forward(x, "E4M3_RNE", x.numel(), False, 1.0)
            hip.gemm_(c, a, a)
        names = {e.name for e in prof.events()}
    finally:
        hip.set_tracing(False)
    assert "fpemu::forward" in names and "cmodel::simple_gemm" in names


def test_graphed_forward_of_a_hooked_model():
    """mpemu.graphs.GraphedForward: a whole E4M3-hooked forward replayed as one graph gives the bits of the
    eager hooked forward, on the example input and on new data."""
    from mpemu import mpt_emu
    from mpemu.graphs import GraphedForward

    torch.manual_seed(1)
    model = torch.nn.Sequential(torch.nn.Linear(96, 192), torch.nn.GELU(), torch.nn.Linear(192, 96),
                                torch.nn.LayerNorm(96)).to(DEV).eval()
    model, _ = mpt_emu.quantize_model(model, dtype="e4m3", device="cuda")
    x0 = torch.randn(300, 96, device=DEV)
    fwd = GraphedForward(model, x0)
    for seed in (0, 1, 2):
        x = x0 if seed == 0 else torch.randn(300, 96, device=DEV, generator=torch.Generator(DEV).manual_seed(seed)) * 3
        with torch.no_grad():
            want = model(x).clone()
        got = fwd(x)
        assert torch.equal(want.view(torch.int32), got.view(torch.int32)), seed
    with pytest.raises(ValueError):
        fwd(torch.randn(10, 96, device=DEV))


def test_memoised_quantisation_stays_out_of_graph_capture():
    """A result produced under stream capture holds nothing until the graph is replayed: such calls neither read nor feed
    the one-entry memo of quantize_tensor, and eager calls after the capture compute their own result."""
    from mpemu import qutils

    cfg = qutils.TensorQuantConfig("e4m3", "rne", "per-tensor")
    x = torch.randn(4096, device="cuda:0") * 3
    eager = qutils.quantize_tensor(x, cfg, None, inplace=False).clone()
    qutils._memo.slot = None
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side), qutils.shared_input_memo():
        qutils.quantize_tensor(x, cfg, None, inplace=False)      # warm: workspace allocation outside the capture
        qutils._memo.slot = None
        with torch.cuda.graph(g, stream=side):
            captured = qutils.quantize_tensor(x, cfg, None, inplace=False)
    torch.cuda.current_stream().wait_stream(side)
    assert qutils._memo.slot is None, "a captured call fed the memo"
    with qutils.shared_input_memo():
        after = qutils.quantize_tensor(x, cfg, None, inplace=False)
    assert after is not captured and torch.equal(after, eager)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(captured, eager)
