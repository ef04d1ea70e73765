"""The C-ABI library loads and exports every symbol include/fpemu_hip.h declares; the mode
table agrees with the oracle's.  No compute calls (CPU only)."""
import ctypes
import os
import re

import pytest

import orc
from _util import ALL_MODES

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "fpemu_hip.h")
SO = os.path.join(ROOT, "fp8-emulation-toolkit_amd", "mpemu", "pytquant", "hip", "libfpemu_hip.so")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(fpemu_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    names = declared_functions()
    assert len(names) >= 9, names
    L = ctypes.CDLL(SO)
    for n in names:
        assert hasattr(L, n), "libfpemu_hip.so does not export " + n


def test_header_cites_reference_for_each_entry_point():
    src = open(HEADER).read()
    for token in ("hip/fpemu_impl.cpp:27-53", "cpp/fpemu_impl.cpp:1878-2000", "qutils.py:283",
                  "qutils.py:284-285"):
        assert token in src, token


def test_mode_table_matches_oracle():
    from mpemu.pytquant.hip import _binding as b

    for m in ALL_MODES:
        assert b.parse_mode(m) == orc.parse_mode(m), m
    with pytest.raises(ValueError):
        b.parse_mode("E4M3_FOO")
    with pytest.raises(ValueError):
        b.parse_mode("")
    assert b.parse_mode("E4M3_IEEE_RNE") == (7, 1) and b.parse_mode("FP4_NEAREST") == (8, 1)


def test_argument_errors_without_gpu():
    """Argument validation happens before any HIP call, so it is testable on a CPU-only box."""
    from mpemu.pytquant.hip import _binding as b

    L = b.lib()
    assert L.fpemu_cast(None, None, 16, 0, 1, 1, 1.0, 0, 0, 0, None) == -1       # null buffers
    assert b"null" in L.fpemu_last_error()
    assert L.fpemu_cast(16, 16, 16, 7, 1, 1, 1.0, 0, 0, 0, None) == -1           # bad dtype
    assert L.fpemu_cast(16, 16, 16, 0, 1, 5, 1.0, 0, 0, 0, None) == -1           # E4M3_RPINF undefined
    assert L.fpemu_quantize_blocks(16, 16, 100, 7, 0, 1, 1, 0, 0, None) == -1    # 100 % 7 != 0
    assert L.fpemu_quantize_blocks(16, 16, 64, 8, 0, 3, 1, 0, 0, None) == -1     # BF16 has no block norm
    assert L.fpemu_quantize_blocks(16, 16, 64, 8, 0, 5, 1, 0, 0, None) == -1     # nor E5M2_DAZ
    assert L.fpemu_gemm_f32(None, 16, 16, 4, 4, 4, 4, 4, 4, 1.0, 0, 0, 1, None) == -1
    assert L.fpemu_gemm_f32(16, 16, 16, 4, 4, 4, 4, 4, 4, 1.0, 0, 0, 9, None) == -1   # unknown variant
    assert L.fpemu_gemm_f32(18, 16, 16, 4, 4, 4, 4, 4, 4, 1.0, 0, 0, 1, None) == -4   # misaligned C
    assert L.fpemu_cast(18, 18, 16, 0, 1, 1, 1.0, 0, 0, 0, None) == -4           # misaligned fp32
    assert L.fpemu_cast(16, 16, 16, 0, 1, 1, float("inf"), 0, 0, 0, None) == -1  # cpp:772-776
    assert b"scale" in L.fpemu_last_error()
    # `scale > FLT_MAX` is the reference's whole check: NaN / -Inf scales are not rejected (they give what the
    # arithmetic gives; tests/test_gpu_qutils.py runs one).  n = 0 here: nothing is launched without a GPU.
    assert L.fpemu_cast(16, 16, 0, 0, 1, 1, float("-inf"), 0, 0, 0, None) == 0
    assert L.fpemu_cast(16, 16, 0, 0, 1, 1, float("nan"), 0, 0, 0, None) == 0
    assert L.fpemu_forward(16, 16, 0, b"E4M3_RNE", 16, float("inf"), 0, 1, 0, 0, None) == -1
    assert L.fpemu_quantize_per_tensor_static(16, 16, (1 << 36) + 1, 0, 1, 1, 448.0, 1, None, 0, 0, None) == -1
    assert L.fpemu_quantize_per_tensor_static(16, 16, 0, 0, 1, 1, 448.0, 1, None, 0, 0, None) == 0   # empty
    assert L.fpemu_cast(16, 16, 0, 0, 1, 1, 1.0, 0, 0, 0, None) == 0             # empty: no launch
    assert L.fpemu_forward(16, 16, 0, b"NOPE", 16, 1.0, 0, 1, 0, 0, None) == -2
    assert L.fpemu_forward(None, None, 0, b"E4M3_RNE", 16, 1.0, 1, 4, 0, 0, None) == -1  # block_norm: null buffers
    assert b"quantize_blocks" in L.fpemu_last_error()
    assert L.fpemu_amax(None, 4, 0, None, None) == -1
    tab = (b._TensorEntry * 2)()
    tab[0].inp, tab[0].out, tab[0].n = 16, 16, 0
    tab[1].inp, tab[1].out, tab[1].n = None, None, 4
    tp = ctypes.cast(tab, ctypes.c_void_p)
    assert L.fpemu_cast_multi(tp, 0, 0, 1, 1, 1.0, 0, 0, None) == 0              # empty list
    assert L.fpemu_cast_multi(tp, 1, 0, 1, 1, 1.0, 0, 0, None) == 0              # one empty tensor: no launch
    assert L.fpemu_cast_multi(tp, 2, 0, 1, 1, 1.0, 0, 0, None) == -1             # null buffer with n > 0
    assert L.fpemu_cast_multi(None, 2, 0, 1, 1, 1.0, 0, 0, None) == -1
    assert L.fpemu_cast_multi(tp, 1, 0, 1, 5, 1.0, 0, 0, None) == -1             # undefined mode pair
    assert L.fpemu_cast_multi(tp, 1, 0, 1, 1, float("inf"), 0, 0, None) == -1   # scale > FLT_MAX
    assert L.fpemu_cast_multi(tp, 1, 0, 1, 1, float("nan"), 0, 0, None) == 0    # not rejected (one empty tensor)
    # the group-aware halves (ABI 9)
    assert L.fpemu_amax_multi(tp, 0, 0, None, None) == 0
    assert L.fpemu_amax_multi(tp, 1, 0, None, None) == -1                        # null words
    assert L.fpemu_amax_multi(tp, 2, 0, 16, None) == -1                          # null buffer with n > 0
    assert L.fpemu_cast_multi_scaled(tp, 1, 0, 3, 1, 16, 1.0, 1, 0, 0, None) == -1   # BF16 takes no scale
    assert L.fpemu_cast_multi_scaled(tp, 2, 0, 1, 1, 16, 448.0, 1, 0, 0, None) == -1 # null buffer with n > 0
    assert L.fpemu_amax_rows(16, 0, 8, 0, None, None) == 0                       # no rows
    assert L.fpemu_amax_rows(16, 4, 8, 0, None, None) == -1                      # null words
    assert L.fpemu_amax_rows(None, 4, 8, 0, 16, None) == -1
    assert L.fpemu_amax_rows(18, 4, 8, 0, 16, None) == -4
    assert L.fpemu_cast_rows_amax(16, 16, 4, 8, 0, 1, 1, None, 448.0, 1, 0, 0, 0, None) == -1
    assert L.fpemu_cast_rows_amax(16, 16, 4, 0, 0, 1, 1, 16, 448.0, 1, 0, 0, 0, None) == 0    # empty rows
    tab[1].inp, tab[1].out = 18, 18
    assert L.fpemu_cast_multi(tp, 2, 0, 1, 1, 1.0, 0, 0, None) == -4             # misaligned fp32
    assert L.fpemu_amax(16, 0, 0, 16, None) == 0
    assert L.fpemu_cast_scaled(16, 16, 4, 0, 1, 1, None, 448.0, 1, 0, 0, 0, None) == -1
    assert L.fpemu_quantize_per_tensor(16, 16, 4, 0, 1, 1, 448.0, 1, None, 0, 0, None) == -1


def test_header_is_plain_c_and_cxx():
    """include/fpemu_hip.h is the drop-in boundary: it must compile as C99 and as C++ on its own (no torch, no HIP
    headers), so a cgo / JNI / pybind / ctypes-generating consumer can include it as is."""
    import shutil
    import subprocess

    hdr = os.path.join(ROOT, "include", "fpemu_hip.h")
    for cc, args in (("gcc", ["-std=c99", "-pedantic", "-x", "c"]), ("g++", ["-std=c++11", "-x", "c++"])):
        if shutil.which(cc) is None:
            pytest.skip(cc + " not installed")
        r = subprocess.run([cc, "-Wall", "-Werror", "-fsyntax-only"] + args + [hdr], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr


def test_weight_update_operand_rule_without_gpu():
    """Which weight updates read the input planes in place of a gathered operand (fpemu_gemm_wu_implicit_ok_: host
    logic only): stride-1 shapes on the small-tile chain kernels whose rows of pixels are whole 16-byte vectors."""
    L = ctypes.CDLL(SO)
    ok = L.fpemu_gemm_wu_implicit_ok_
    ok.restype = ctypes.c_int
    ok.argtypes = [ctypes.c_void_p] + [ctypes.c_int64] * 5 + [ctypes.c_int]

    def rule(K, C, R, S, OH, OW, ptr=4096, variant=1):
        return ok(ptr, K, C * R * S, OH * OW, OW, K * OH * OW, variant)

    assert rule(64, 64, 3, 3, 56, 56) == 1         # ResNet-50 layer1 3 x 3 (16 x 16 tiles)
    assert rule(128, 128, 3, 3, 28, 28) == 1       # layer2 3 x 3 (32 x 32 tiles)
    assert rule(64, 256, 1, 1, 56, 56) == 1        # 1 x 1
    assert rule(256, 256, 3, 3, 14, 14) == 0       # 14 pixels per row: not whole vectors (and the large-tile kernel)
    assert rule(512, 512, 3, 3, 8, 8) == 0         # 64 x 64 tiles: the gathered form
    assert rule(64, 64, 3, 3, 56, 56, ptr=4100) == 0   # gout not 16-byte aligned
    assert rule(64, 64, 3, 3, 56, 56, variant=0) == 0  # the separate-rounding variant runs on the VALU kernels
    assert rule(64, 64, 3, 3, 1024, 1024) == 0     # pixel index past the exact range of the reciprocal multiply
