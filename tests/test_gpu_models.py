"""Module hooks and emulators on the GPU (BASELINE configs 3 and 4 in miniature) against outputs
of the REFERENCE's mpt_emu.quantize_model run on CPU (tests/golden/models.npz)."""
import copy

import numpy as np
import pytest
import torch

import orc
from _util import assert_bits_equal, npz

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _f(a):
    return torch.from_numpy(a.view(np.float32).copy())


def _mlp(g):
    m = torch.nn.Sequential(torch.nn.Linear(40, 64), torch.nn.ReLU(), torch.nn.Linear(64, 24))
    m.load_state_dict({k.split("/", 1)[1]: _f(g[k]) for k in g.files if k.startswith("mlp_init/")})
    return m


def _conv(g):
    m = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.BatchNorm2d(8), torch.nn.ReLU(),
                            torch.nn.Conv2d(8, 6, 3))
    sd = {k.split("/", 1)[1]: _f(g[k]) for k in g.files if k.startswith("conv_init/")}
    m.load_state_dict(sd, strict=False)
    return m


@pytest.mark.parametrize("dtype", ["e4m3", "e5m2", "e3m4", "hybrid"])
@pytest.mark.parametrize("calibrate", [False, True])
def test_quantize_model_mlp_vs_reference(dtype, calibrate):
    from mpemu import mpt_emu

    g = npz("models.npz")
    m = _mlp(g).to(DEV).eval()
    m, emu = mpt_emu.quantize_model(m, dtype=dtype, calibrate=calibrate, device="cuda")
    seen = {}
    m[0].register_forward_hook(lambda mod, i, o: seen.__setitem__("in0", i[0].detach().clone()))
    m[2].register_forward_hook(lambda mod, i, o: seen.__setitem__("in2", i[0].detach().clone()))
    with torch.no_grad():
        y = m(_f(g["x_mlp"]).to(DEV))
    tag = "mlp/%s/%d/" % (dtype, int(calibrate))
    # exact pins: quantised weights, and the quantised tensor entering the first layer
    assert_bits_equal(m[0].weight.detach().cpu().numpy(), g[tag + "w0"], tag + "w0")
    assert_bits_equal(m[2].weight.detach().cpu().numpy(), g[tag + "w2"], tag + "w2")
    assert_bits_equal(seen["in0"].cpu().numpy(), g[tag + "in0"], tag + "in0")
    # downstream of a GEMM (host BLAS there, hipBLAS here): same up to one quantum of the format
    ref_in2, ref_y = _f(g[tag + "in2"]), _f(g[tag + "y"])
    assert torch.allclose(seen["in2"].cpu(), ref_in2, rtol=0.26, atol=2e-2)
    # (not compared exactly: the per-tensor scale comes from the amax of a GEMM output whose last
    # bits differ between BLAS libraries, which moves every dequantised value by an ulp)
    assert torch.allclose(y.cpu(), ref_y, rtol=0.1, atol=0.15)


def test_scale_shift_and_conv_vs_reference():
    from mpemu import mpt_emu
    from mpemu.scale_shift import ScaleShift, replace_batchnorms_with_scaleshifts

    g = npz("models.npz")
    c = replace_batchnorms_with_scaleshifts(_conv(g).eval())
    assert isinstance(c[1], ScaleShift) and repr(c[1]) == "ScaleShift(8)"
    # host-side torch arithmetic (sqrt / div on the CPU of whichever box runs this): 1-ulp differences
    # between hosts were observed, so the bit-exact pin lives in tests/test_host_logic.py (same host
    # as the fixture generator) and this one allows an ulp
    assert torch.allclose(c[1].weight.detach(), _f(g["conv/ss_weight"]), rtol=3e-7, atol=0)
    assert torch.allclose(c[1].bias.detach(), _f(g["conv/ss_bias"]), rtol=1e-6, atol=1e-7)
    x = _f(g["x_conv"])
    with torch.no_grad():
        y_ss = c(x)
    assert torch.allclose(y_ss, _f(g["conv/y_ss"]), rtol=1e-5, atol=1e-5)
    assert torch.allclose(y_ss, _f(g["conv/y_bn"]), rtol=1e-4, atol=1e-4)   # == BatchNorm in eval mode
    c = c.to(DEV)
    c, emu = mpt_emu.quantize_model(c, dtype="e4m3", device="cuda")
    assert_bits_equal(c[0].weight.detach().cpu().numpy(), g["conv/e4m3/w0"], "conv w0")
    assert_bits_equal(c[3].weight.detach().cpu().numpy(), g["conv/e4m3/w3"], "conv w3")
    with torch.no_grad():
        y = c(x.to(DEV))
    assert torch.allclose(y.cpu(), _f(g["conv/e4m3/y"]), rtol=0.1, atol=0.1)


def test_quantize_model_fuse_bn_and_errors():
    from mpemu import mpt_emu
    from mpemu.scale_shift import ScaleShift

    g = npz("models.npz")
    m, emu = mpt_emu.quantize_model(_conv(g).eval().to(DEV), dtype="e4m3", fuse_bn=True, device="cuda")
    assert isinstance(m[1], ScaleShift)
    with torch.no_grad():
        assert torch.isfinite(m(_f(g["x_conv"]).to(DEV))).all()
    with pytest.raises(RuntimeError):
        mpt_emu.quantize_model(_mlp(g).to(DEV), dtype="e4m3_rne", device="cuda")   # README typo'd dtype
    with pytest.raises(RuntimeError):
        mpt_emu.quantize_model(_mlp(g), dtype="e4m3", device="cpu")
    with pytest.raises(RuntimeError):
        mpt_emu.quantize_model(_mlp(g).to(DEV), dtype="e4m3", hw_patch="bogus", device="cuda")


def _is_fixed_point(t, mode, scale=1.0):
    x = t.detach().float().cpu().numpy()
    return np.array_equal(orc.cast(x, mode, scale=scale).view(np.uint32), x.view(np.uint32))


def test_hybrid_training_step():
    """Config 4 in miniature: E4M3 per-tensor weights / activations forward, E5M2 stochastic
    gradients backward, FLOAT16_STOCHASTIC master weights after the optimizer step."""
    from mpemu import mpt_emu
    from mpemu.pytquant.hip import fpemu as F

    F.manual_seed(7)
    g = npz("models.npz")
    model = _mlp(g).to(DEV).train()
    w_before = model[0].weight.detach().clone()
    opt = torch.optim.SGD(model.parameters(), lr=0.05)
    model, emu = mpt_emu.initialize(model, opt, training_algo="hybrid", device="cuda")
    assert emu.emulator.is_training and emu.emulator.use_fp16_master
    x = _f(g["x_mlp"]).to(DEV).requires_grad_(True)
    y = model(x)
    # forward pre-hook quantised the weights in place with the per-tensor max scale
    wn = w_before.cpu().numpy()
    s = orc.scale_from_amax(orc.amax(wn), 448.0)
    assert_bits_equal(model[0].weight.detach().cpu().numpy(), orc.cast(wn, "E4M3_RNE", scale=s), "hooked weight")
    yn = y.detach().cpu().numpy()
    sy = orc.scale_from_amax(orc.amax(yn), 448.0)
    del sy  # the output was cast in place with the scale of its pre-cast amax: nothing exact to pin here
    y.square().mean().backward()
    for p in (model[0].weight, model[2].weight):
        assert p.grad is not None and torch.isfinite(p.grad).all()
        assert _is_fixed_point(p.grad, "E5M2_RTZ"), "weight gradients are E5M2 values"
    assert _is_fixed_point(x.grad, "E5M2_RTZ"), "input gradients are E5M2 values"
    emu.optimizer_step(opt)
    assert emu.emulator.global_steps == 1
    for p in model.parameters():
        assert _is_fixed_point(p, "FLOAT16_RNE"), "master parameters are fp16 values after the step"
    assert not torch.equal(model[0].weight.detach(), w_before)


def test_direct_e5m2_training_and_module_lists():
    from mpemu import mpt_emu
    from mpemu.module_wrappers import EltwiseAdd, Matmul

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a = torch.nn.Linear(16, 16)
            self.b = torch.nn.Linear(16, 16)
            self.mm = Matmul()
            self.add = EltwiseAdd()
            self.ln = torch.nn.LayerNorm(16)

        def forward(self, x):
            return self.ln(self.add(self.mm(self.a(x), self.b(x).t()), x))

    torch.manual_seed(3)
    net = Net().to(DEV).train()
    opt = torch.optim.SGD(net.parameters(), lr=0.01)
    net, emu = mpt_emu.initialize(net, opt, training_algo="direct", device="cuda",
                                  list_exempt_layers=["b"], list_layers_output_fused=["mm"])
    d = emu.emulator.model_qconfig_dict
    assert set(d) == {"a", "mm", "add", "ln"}
    assert d["mm"].wt_qconfig is None and d["mm"].oact_qconfig is None     # no weight; output fused
    assert d["ln"].wt_qconfig is None and d["a"].wt_qconfig is not None
    x = torch.randn(16, 16, device=DEV)
    net(x).sum().backward()
    emu.optimizer_step(opt)
    emu.blacklist_modules([torch.nn.LayerNorm])
    assert "ln" not in emu.emulator.model_qconfig_dict
    emu.whitelist_modules([torch.nn.LayerNorm])
    assert "ln" in emu.emulator.model_qconfig_dict
    net(x).sum().backward()


def test_master_param_update_equals_the_reference_loop():
    """update_fp16_master_params batches the FLOAT16_STOCHASTIC casts between embedding parameters into
    multi-tensor launches: same bits and same call counter as the reference's per-parameter loop
    (hybrid_emu.py:110-134)."""
    from mpemu import mpt_emu
    from mpemu.pytquant import fpemu_cuda
    from mpemu.pytquant.hip import _binding as hip

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a = torch.nn.Linear(40, 70)
            self.embeddings = torch.nn.Embedding(50, 16)
            self.b = torch.nn.Linear(70, 33)
            self.c = torch.nn.Conv2d(3, 8, 3)
            self.norm = torch.nn.LayerNorm(33)

    torch.manual_seed(2)
    model = Net().to(DEV)
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    model, emu = mpt_emu.initialize(model, opt, training_algo="hybrid", device="cuda")
    emu.set_master_param_precision("fp16")
    emu.set_embedding_precision("e4m3", emb_norm=True)
    before = [p.detach().clone() for p in model.parameters()]
    names = [n for n, _ in model.named_parameters()]

    hip.set_rng_state((3, 40))
    want = []
    for t, name in zip(before, names):       # the reference's loop, one call per cast
        t = fpemu_cuda.FPEmuOp.apply(t.clone(), "FLOAT16_STOCHASTIC", True)
        if "embeddings" in name:
            t = fpemu_cuda.FPEmuOp.apply(t, "E4M3_STOCHASTIC", True, True, t.size()[1])
        want.append(t)
    state = hip.get_rng_state()

    hip.set_rng_state((3, 40))
    emu.emulator.update_master_params(opt)
    assert hip.get_rng_state() == state
    for p, w, name in zip(model.parameters(), want, names):
        assert torch.equal(p.detach().view(torch.int32), w.view(torch.int32)), name
        assert _is_fixed_point(p, "FLOAT16_RNE"), name


def test_bert_base_ptq_e4m3():
    """BASELINE config 3, synthetic: random-init BERT-base, E4M3 PTQ with per-channel weights and
    per-tensor input activations through the module hooks."""
    transformers = pytest.importorskip("transformers")
    from mpemu import mpt_emu

    torch.manual_seed(0)
    cfg = transformers.BertConfig()
    model = transformers.BertModel(cfg).to(DEV).eval()
    lin = model.encoder.layer[3].attention.self.query
    w0 = lin.weight.detach().cpu().numpy().copy()
    emb0 = model.embeddings.word_embeddings.weight.detach().cpu().numpy()[:64].copy()
    model, emu = mpt_emu.quantize_model(model, dtype="e4m3", device="cuda",
                                        list_exempt_layers=["pooler.dense"])
    assert "pooler.dense" not in emu.emulator.model_qconfig_dict
    got = lin.weight.detach().cpu().numpy()
    for r in (0, 1, 17, 767):
        s = orc.scale_from_amax(orc.amax(w0[r]), 448.0)
        assert_bits_equal(got[r], orc.cast(w0[r], "E4M3_RNE", scale=s), "bert query row %d" % r)
    e = model.embeddings.word_embeddings.weight.detach().cpu().numpy()
    for r in (0, 5, 63):
        s = orc.scale_from_amax(orc.amax(emb0[r]), 448.0)
        assert_bits_equal(e[r], orc.cast(emb0[r], "E4M3_RNE", scale=s), "bert embedding row %d" % r)
    ids = torch.randint(0, cfg.vocab_size, (2, 128), device=DEV)
    with torch.no_grad():
        out = model(input_ids=ids).last_hidden_state
    assert out.shape == (2, 128, 768) and torch.isfinite(out).all()


def test_resnet50_hybrid_training_step():
    """BASELINE config 4 at the real model's shape: ResNet-50 (random init, batch 8 of 3x64x64), one
    mixed-precision training step through mpt_emu.initialize(training_algo="hybrid"): E4M3 per-tensor
    weights / activations forward, E5M2 stochastic gradients, fp16 stochastic master weights."""
    from _resnet import ResNet50

    from mpemu import mpt_emu
    from mpemu.pytquant.hip import fpemu as F

    F.manual_seed(11)
    torch.manual_seed(0)
    model = ResNet50(classes=100).to(DEV).train()
    opt = torch.optim.SGD(model.parameters(), lr=0.01, momentum=0.9)
    model, emu = mpt_emu.initialize(model, opt, training_algo="hybrid", device="cuda")
    d = emu.emulator.model_qconfig_dict
    convs = [n for n, m in model.named_modules() if isinstance(m, torch.nn.Conv2d)]
    assert len(convs) == 53 and all(n in d for n in convs) and "fc" in d
    x = torch.randn(8, 3, 64, 64, device=DEV)
    target = torch.randint(0, 100, (8,), device=DEV)
    w0 = model.layers[3].conv2.weight.detach().clone()
    loss = torch.nn.functional.cross_entropy(model(x), target)
    assert torch.isfinite(loss)
    # the hooked conv weight now holds E4M3 values under its per-tensor max scale
    wn = w0.cpu().numpy()
    s = orc.scale_from_amax(orc.amax(wn), 448.0)
    assert_bits_equal(model.layers[3].conv2.weight.detach().cpu().numpy(), orc.cast(wn, "E4M3_RNE", scale=s),
                      "ResNet conv weight after the forward pre-hook")
    loss.backward()
    for name in ("layers.3.conv2", "layers.10.conv3", "fc"):
        gw = dict(model.named_modules())[name].weight.grad
        assert gw is not None and torch.isfinite(gw).all() and _is_fixed_point(gw, "E5M2_RTZ"), name
    emu.optimizer_step(opt)
    assert all(_is_fixed_point(p, "FLOAT16_RNE") for p in model.parameters())
    assert not torch.equal(model.layers[3].conv2.weight.detach(), w0)
