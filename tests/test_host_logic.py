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
