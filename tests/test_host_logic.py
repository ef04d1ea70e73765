"""Host-side mirror of the reference's Python interface (no GPU needed)."""
import pytest
import torch

import mpemu
from mpemu import qutils
from mpemu.pytquant import fpemu_cpp, fpemu_cuda


def test_import_surface_matches_reference():
    # mpemu/pytquant/__init__.py:4-13 of the reference: modules with an FPEmuOp each
    assert hasattr(fpemu_cuda, "FPEmuOp") and hasattr(fpemu_cpp, "FPEmuOp")
    assert callable(qutils.fpemu_device_fn) and callable(qutils.quantize_tensor)
    assert mpemu.pytquant.fpemu_cuda is fpemu_cuda


def test_tensor_quant_config_scaling_tokens():
    # qutils.py:23-38 of the reference
    c = qutils.TensorQuantConfig("e4m3", "rne", "per-tensor")
    assert c.scaling and not c.per_channel and c.scaling_method == "max" and not c.fine_grained
    c = qutils.TensorQuantConfig("e4m3", "rne", "per-channel")
    assert not c.scaling and c.per_channel and c.scaling_method == "max"
    c = qutils.TensorQuantConfig("e4m3", "rne", "per-tensor-mean")
    assert c.scaling and c.scaling_method == "mean"
    c = qutils.TensorQuantConfig("e3m4", "rne", "fine-grained", group_size=8)
    assert c.fine_grained and c.group_size == 8 and not c.scaling and not c.per_channel
    c = qutils.TensorQuantConfig("e5m2", "stochastic")
    assert not c.scaling and not c.per_channel and c.group_size == 1
    assert qutils.TensorQuantConfig("int8", "sym_full").scaling_method == "none"
    assert repr(qutils.TensorQuantConfig("e4m3", "rne", "per-channel")) == \
        "[e4m3_rne, scale: per-channel, method: max]"


def test_flt_max_min_table():
    # qutils.py:100-128
    T = qutils.TensorQuantConfig
    assert T("e5m2", "rne").get_flt_max() == 57344.0
    assert T("e4m3", "rne").get_flt_max() == 448.0
    assert T("e4m3", "ieee_rne").get_flt_max() == 240.0
    assert T("e3m4", "rne").get_flt_max() == 30.0
    assert T("fp4", "nearest").get_flt_max() == 1.0
    assert T("e5m2", "rne").get_flt_min() == 1.5258789E-05
    assert T("e4m3", "rne").get_flt_min() == 1.9531250E-03
    assert T("e3m4", "rne").get_flt_min() == 1.5625000E-02


def test_validity_checks():
    with pytest.raises(AssertionError):
        qutils.TensorQuantConfig("e4m3", "rnaz")
    with pytest.raises(AssertionError):
        qutils.TensorQuantConfig("e9m9", "rne")
    qutils.TensorQuantConfig("float16", "rne")  # the reference raises AttributeError here (typo)


def test_none_mode_is_passthrough_and_cpu_tensor_raises():
    x = torch.randn(8)
    assert fpemu_cuda.FPEmuOp.apply(x, "NONE") is x or torch.equal(fpemu_cuda.FPEmuOp.apply(x, "NONE"), x)
    with pytest.raises(RuntimeError, match="ROCm"):
        qutils.fpemu_device_fn(x, "E4M3_RNE", inplace=False)
    with pytest.raises(RuntimeError):
        fpemu_cuda.FPEmuOp.apply(x, "E4M3_RNE")


def test_integer_modes_stay_in_torch():
    x = torch.linspace(-1, 1, 64)
    q = qutils.fpemu_device_fn(x, "INT8", inplace=False)
    assert q.shape == x.shape and (q - x).abs().max() < 2.0 / 255 + 1e-6


def test_module_quant_config_repr():
    m = qutils.ModuleQuantConfig(wt_qconfig=qutils.TensorQuantConfig("e4m3", "rne"))
    assert repr(m).startswith("[weights: [e4m3_rne")
    assert m.device == "cuda" and m.patch_impl == "None"


def test_scale_shift_matches_reference_fixture():
    """BatchNorm2d -> ScaleShift folding against the reference's scale_shift.py output."""
    import numpy as np

    from _util import npz
    from mpemu.scale_shift import ScaleShift, replace_batchnorms_with_scaleshifts

    g = npz("models.npz")
    f = lambda a: torch.from_numpy(a.view(np.float32).copy())
    m = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.BatchNorm2d(8), torch.nn.ReLU(),
                            torch.nn.Conv2d(8, 6, 3))
    m.load_state_dict({k.split("/", 1)[1]: f(g[k]) for k in g.files if k.startswith("conv_init/")}, strict=False)
    m = replace_batchnorms_with_scaleshifts(m.eval())
    assert isinstance(m[1], ScaleShift)
    assert torch.allclose(m[1].weight.detach(), f(g["conv/ss_weight"]), rtol=3e-7, atol=0)
    assert torch.allclose(m[1].bias.detach(), f(g["conv/ss_bias"]), rtol=1e-6, atol=1e-7)
    with torch.no_grad():
        y = m(f(g["x_conv"]))
    assert torch.allclose(y, f(g["conv/y_bn"]), rtol=1e-4, atol=1e-4)


def test_emulator_recipe_tables():
    """Default TensorQuantConfigs per recipe (reference e5m2_emu.py:55-61, e4m3_emu.py:46-49,176-185,
    e3m4_emu.py:45-48,159-168, hybrid_emu.py:56-61,400-413)."""
    from mpemu.e3m4_emu import E3M4Emulator
    from mpemu.e4m3_emu import E4M3Emulator
    from mpemu.e5m2_emu import E5M2Emulator
    from mpemu.hybrid_emu import HybridEmulator

    m = torch.nn.Linear(4, 4)
    h = HybridEmulator(m, None)
    assert repr(h.wt_qconfig) == "[e4m3_rne, scale: per-tensor, method: max]"
    assert repr(h.wtgrad_qconfig) == "[e5m2_stochastic, scale: None, method: max]" and h.emb_qconfig is None
    h.set_default_inference_qconfig()
    assert repr(h.wt_qconfig) == "[e3m4_rne, scale: per-channel, method: max]" and h.igrad_qconfig is None
    e5 = E5M2Emulator(m, None)
    assert repr(e5.oact_qconfig) == "[e5m2_rne, scale: None, method: max]"
    e5.set_default_inference_qconfig()
    assert repr(e5.wt_qconfig) == "[e5m2_rne, scale: per-channel, method: mean]" and e5.oact_qconfig is None
    e4 = E4M3Emulator(m, None)
    assert repr(e4.wt_qconfig) == "[e4m3_rne, scale: None, method: max]" and not e4.is_training
    e4.set_default_inference_qconfig()
    assert repr(e4.iact_qconfig) == "[e4m3_rne, scale: per-tensor, method: max]"
    e3 = E3M4Emulator(m, None)
    assert repr(e3.wt_qconfig) == "[e3m4_rne, scale: per-channel, method: max]"
    e3.set_calibration_qconfig()
    assert repr(e3.wt_qconfig) == "[e3m4_rne, scale: per-tensor, method: max]"
    assert torch.nn.LayerNorm in h.whitelist and torch.nn.LayerNorm not in e4.whitelist
