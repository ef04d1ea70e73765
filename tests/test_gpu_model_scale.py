"""BASELINE configs 3 and 4 at MODEL SCALE, tensor by tensor against the oracle.

Config 3: a BERT-base-shaped encoder (vocab 30522, d 768, 12 heads, ff 3072, 12 layers, seq 128, batch 8) whose
attention products and residual adds are `module_wrappers` modules -- as in the reference's own config-3 model
(examples/inference/bert/modeling_bert.py:251-252, 370, 450) -- through `mpt_emu.quantize_model(dtype="e4m3")`:
every Linear / Embedding weight row against the per-row oracle, and the tensors that hooked modules RECEIVE against
the oracle's cast of the tensors that went in (reference qutils.py:478-534).

Config 4: ResNet-50, batch 32 at 224 x 224, one hybrid training step (`mpt_emu.initialize(training_algo="hybrid")`,
reference hybrid_emu.py:56-61, 110-191; qutils.py:586-614): all 54 hooked weights, stochastic E5M2 gradients of
chosen layers at the (seed, offset) their calls drew, and every master parameter after the optimizer step against
the oracle's FLOAT16_STOCHASTIC at the offsets the update consumed.  The Philox stream is a pure function of
(seed, offset, element), so the stochastic tensors are compared bit for bit."""
import numpy as np
import pytest
import torch

import orc
from _util import assert_bits_equal

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rows_oracle(w, mode, flt_max):
    """Per-channel "max" policy of quantize_tensor (reference qutils.py:289-311): row r of the [rows, cols] view
    cast with scale_from_amax(max|row r|)."""
    w2 = np.ascontiguousarray(w, dtype=np.float32).reshape(w.shape[0], -1)
    out = np.empty_like(w2)
    for r in range(w2.shape[0]):
        out[r] = orc.cast(w2[r], mode, scale=orc.scale_from_amax(orc.amax(w2[r]), flt_max))
    return out.reshape(w.shape)


def _per_tensor_oracle(x, mode, flt_max, **kw):
    return orc.cast(x, mode, scale=orc.scale_from_amax(orc.amax(x), flt_max), **kw)


def test_config3_bert_base_shape_through_module_wrappers():
    from _zoo import TinyBert

    from mpemu import mpt_emu

    torch.manual_seed(0)
    model = TinyBert(vocab=30522, d=768, heads=12, ff=3072, layers=12, seq=128).to(DEV).eval()
    before = {n: m.weight.detach().cpu().numpy().copy() for n, m in model.named_modules()
              if isinstance(m, (torch.nn.Linear, torch.nn.Embedding))}
    assert len(before) == 12 * 6 + 1
    model, emu = mpt_emu.quantize_model(model, dtype="e4m3", device="cuda")
    d = emu.emulator.model_qconfig_dict
    mods = dict(model.named_modules())

    # every weight row of every Linear / Embedding: per-channel E4M3_RNE (e4m3_emu.py:181-185)
    for name, w0 in before.items():
        assert name in d, name
        assert_bits_equal(mods[name].weight.detach().cpu().numpy(), _rows_oracle(w0, "E4M3_RNE", 448.0),
                          "config 3 weight %s" % name)

    # hooked activations: a plain pre-hook AHEAD of the quantisation hook sees what goes in, one BEHIND it sees
    # what the module receives
    watch = ["layers.0.q", "layers.0.scores", "layers.0.context", "layers.0.ff1", "layers.11.res2", "pool"]
    raw, got = {}, {}
    for name in watch:
        m = mods[name]
        m.register_forward_pre_hook(lambda mod, inp, name=name: raw.__setitem__(name, [t.detach().clone() for t in inp]),
                                    prepend=True)
        m.register_forward_pre_hook(lambda mod, inp, name=name: got.__setitem__(name, [t.detach().clone() for t in inp]))
    ids = torch.randint(0, 30522, (8, 128), device=DEV, generator=torch.Generator(device=DEV).manual_seed(5))
    with torch.no_grad():
        out = model(ids)
    assert out.shape == (8, 128, 2) and torch.isfinite(out).all()
    ninputs = {"layers.0.q": 1, "layers.0.scores": 2, "layers.0.context": 2, "layers.0.ff1": 1, "layers.11.res2": 2,
               "pool": 2}
    for name in watch:
        assert len(raw[name]) == len(got[name]) == ninputs[name], name
        for i, (a, b) in enumerate(zip(raw[name], got[name])):
            assert a.data_ptr() != b.data_ptr(), "inputs are quantised out of place"
            an = a.cpu().numpy()
            assert_bits_equal(b.cpu().numpy(), _per_tensor_oracle(an, "E4M3_RNE", 448.0),
                              "config 3 input %d of %s" % (i, name))
            assert not np.array_equal(an, b.cpu().numpy()), "the hook did quantise %s" % name
    # outputs are not quantised in this preset (oact None) and the weights were quantised once, up front
    for name, w0 in before.items():
        if name.startswith("layers.11."):
            assert_bits_equal(mods[name].weight.detach().cpu().numpy(), _rows_oracle(w0, "E4M3_RNE", 448.0),
                              "config 3 weight %s after the forward pass" % name)


class _Recorder(object):
    """Stands in for qutils.quantize_tensor during one training step: forwards every call to the real function
    and records, per call, the stochastic stream position it drew from; for the watched configurations also the
    tensor that went in and the tensor that came out."""

    def __init__(self, real, hip, watched):
        self.real, self.hip, self.watched = real, hip, watched
        self.calls = []       # (key or None, offset_before, offset_after)
        self.tensors = {}     # key -> (input clone, output clone, seed, offset)

    def __call__(self, tensor, qtconfig, qtparams, inplace=False):
        key = self.watched.get(id(qtconfig))
        seed, off = self.hip.get_rng_state()
        src = tensor.detach().clone() if key is not None else None
        out = self.real(tensor, qtconfig, qtparams, inplace=inplace)
        self.calls.append((key, off, self.hip.get_rng_state()[1]))
        if key is not None:
            self.tensors.setdefault(key, []).append((src, out.detach().clone(), seed, off))
        return out


def test_config4_resnet50_batch32_224_hybrid_step(monkeypatch):
    from _resnet import ResNet50

    from mpemu import mpt_emu, qutils
    from mpemu.pytquant.hip import _binding as hip
    from mpemu.pytquant.hip import fpemu as F

    F.manual_seed(1234)
    torch.manual_seed(0)
    model = ResNet50(classes=1000).to(DEV).train()
    opt = torch.optim.SGD(model.parameters(), lr=0.05, momentum=0.9)
    model, emu = mpt_emu.initialize(model, opt, training_algo="hybrid", device="cuda")
    mods = dict(model.named_modules())
    hooked = [n for n, m in mods.items() if isinstance(m, (torch.nn.Conv2d, torch.nn.Linear))]
    assert len(hooked) == 54 and all(n in emu.emulator.model_qconfig_dict for n in hooked)
    w_before = {n: mods[n].weight.detach().cpu().numpy().copy() for n in hooked}

    layers = ("conv1", "layers.3.conv2", "layers.10.conv3", "fc")
    watched = {}
    for n in layers:
        qc = mods[n].qconfig
        watched[id(qc.ograd_qconfig)] = (n, "ograd")
        watched[id(qc.wtgrad_qconfig)] = (n, "wtgrad")
        if n != "conv1":  # the stem's input is the image batch: no gradient flows to it
            watched[id(qc.igrad_qconfig)] = (n, "igrad")
    rec = _Recorder(qutils.quantize_tensor, hip, watched)
    monkeypatch.setattr(qutils, "quantize_tensor", rec)

    g = torch.Generator(device=DEV).manual_seed(9)
    x = torch.randn(32, 3, 224, 224, device=DEV, generator=g)
    target = torch.randint(0, 1000, (32,), device=DEV, generator=g)
    loss = torch.nn.functional.cross_entropy(model(x), target)
    assert torch.isfinite(loss)

    # all 54 hooked weights: per-tensor E4M3_RNE in place by the forward pre-hooks (hybrid_emu.py:56-61)
    for n in hooked:
        assert_bits_equal(mods[n].weight.detach().cpu().numpy(), _per_tensor_oracle(w_before[n], "E4M3_RNE", 448.0),
                          "config 4 weight %s" % n)
    fwd_calls = len(rec.calls)
    assert all(a == b for _, a, b in rec.calls), "the forward pass draws no stochastic noise"

    loss.backward()
    bwd = rec.calls[fwd_calls:]
    # every backward cast is E5M2_STOCHASTIC and consumes exactly one call offset, in order
    assert len(bwd) >= 3 * 53 and [b - a for _, a, b in bwd] == [1] * len(bwd)
    assert [a for _, a, _ in bwd] == list(range(bwd[0][1], bwd[0][1] + len(bwd)))
    for n in layers:
        roles = ("ograd", "wtgrad") + (("igrad",) if n != "conv1" else ())
        for role in roles:
            hits = rec.tensors.get((n, role))
            assert hits and len(hits) == 1, (n, role)
            src, out, seed, off = hits[0]
            sn = src.cpu().numpy()
            assert_bits_equal(out.cpu().numpy(), orc.cast(sn, "E5M2_STOCHASTIC", seed=seed, offset=off),
                              "config 4 %s of %s at offset %d" % (role, n, off))
            assert not np.array_equal(sn, out.cpu().numpy())
    # the weight gradient the optimizer will see IS the hooked one (quantised in place)
    for n in layers:
        assert torch.equal(mods[n].weight.grad, rec.tensors[(n, "wtgrad")][0][1]), n

    # optimizer step + master parameters: FLOAT16_STOCHASTIC, parameter i at offset0 + i (hybrid_emu.py:110-134)
    snap = {}
    real_update = emu.emulator.update_master_params

    def update(optimizer):
        snap["params"] = [p.detach().clone() for p in model.parameters()]
        snap["state"] = hip.get_rng_state()
        real_update(optimizer)

    monkeypatch.setattr(emu.emulator, "update_master_params", update)
    emu.optimizer_step(opt)
    seed, off0 = snap["state"]
    params = list(model.parameters())
    assert len(params) == 161 and hip.get_rng_state() == (seed, off0 + len(params))
    moved = 0
    for i, (p, p0) in enumerate(zip(params, snap["params"])):
        p0n = p0.cpu().numpy()
        assert_bits_equal(p.detach().cpu().numpy(), orc.cast(p0n, "FLOAT16_STOCHASTIC", seed=seed, offset=off0 + i),
                          "config 4 master parameter %d" % i)
        moved += int(not np.array_equal(p0n, p.detach().cpu().numpy()))
    assert moved > 100
