"""The reference-side binding stubs INTEGRATION.md shows (Option B: `fpemu_hip.py`, `simple_gemm_dev.py`,
`simple_conv2d_dev.py`, the modules a maintainer of the reference would drop in place of its compiled extensions) are
taken out of the document, saved next to the library and RUN: they load, their signatures match the C ABI, and what
they return equals the oracle bit for bit."""
import importlib.util
import os
import re
import sys

import numpy as np
import pytest
import torch

import orc
from _util import assert_bits_equal, mixed_inputs

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "fp8-emulation-toolkit_amd", "mpemu", "pytquant", "hip", "libfpemu_hip.so")


def _python_blocks():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    return re.findall(r"```python\n(.*?)```", text, flags=re.S)


def _load(tmp_path, name, source):
    os.symlink(SO, os.path.join(tmp_path, "libfpemu_hip.so")) if not os.path.exists(os.path.join(tmp_path, "libfpemu_hip.so")) else None
    path = os.path.join(tmp_path, name + ".py")
    with open(path, "w") as f:
        f.write(source)
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_option_b_stubs_run_and_match_the_oracle(tmp_path):
    blocks = _python_blocks()
    cast_src = next(b for b in blocks if b.lstrip().startswith("# fpemu_hip.py"))
    cmodel_src = next(b for b in blocks if b.lstrip().startswith("# simple_gemm_dev.py"))
    fpemu_hip = _load(str(tmp_path), "fpemu_hip_doc_stub", cast_src)
    x = mixed_inputs(100003, seed=3)
    t = torch.from_numpy(x).cuda()
    out = fpemu_hip.forward(t, "E4M3_RNE", t.numel(), False, 1.0)[0]          # the reference's call (fpemu.py:54-61)
    assert out is not t
    assert_bits_equal(out.cpu().numpy(), orc.cast(x, "E4M3_RNE"), "doc stub forward")
    out = fpemu_hip.backward(t.clone(), "E5M2_RNE", t.numel(), True, 4.0, False, 1)[0]
    assert_bits_equal(out.cpu().numpy(), orc.cast(x, "E5M2_RNE", scale=4.0), "doc stub backward, in place, scaled")
    h = t.half()
    out = fpemu_hip.forward(h, "E5M2_RNE", h.numel(), False, 1.0)[0]
    assert out.dtype == torch.float16
    with pytest.raises(RuntimeError):
        fpemu_hip.forward(t, "NO_SUCH_MODE", t.numel(), False, 1.0)

    cm = _load(str(tmp_path), "simple_cmodel_doc_stub", cmodel_src)
    rng = np.random.default_rng(5)
    a = rng.standard_normal((70, 90)).astype(np.float32)
    b = rng.standard_normal((90, 50)).astype(np.float32)
    C = torch.zeros(70, 50, device="cuda")
    cm.gemm(C, torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda(), 1.0, False, False)   # simple_gemm.cpp:45-67
    assert_bits_equal(C.cpu().numpy(), orc.gemm(a, b, fused=True), "doc stub gemm")
    xi = rng.standard_normal((2, 8, 10, 10)).astype(np.float32)
    w = rng.standard_normal((16, 8, 3, 3)).astype(np.float32)
    bias = rng.standard_normal(16).astype(np.float32)
    y = torch.empty(2, 16, 5, 5, device="cuda")
    cm.conv2d_fp(y, torch.from_numpy(xi).cuda(), torch.from_numpy(w).cuda(), torch.from_numpy(bias).cuda(), 2, 1, 1, 1)
    assert_bits_equal(y.cpu().numpy(), orc.conv2d_fp(xi, w, bias, 2, 1, True), "doc stub conv2d_fp")
