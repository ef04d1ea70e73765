"""mpemu -- MI355X-native build of the FP8-Emulation-Toolkit's fpemu cast path.

Import surface of the reference package (mpemu/__init__.py) for the cast path:
    mpemu.pytquant.{fpemu_cuda, fpemu_cpp}.FPEmuOp, mpemu.qutils.*
The arithmetic runs in hand-written gfx950 HIP kernels (libfpemu_hip.so, C ABI in
include/fpemu_hip.h); see DESIGN.md.
"""
from . import pytquant
from . import qutils
from . import sharded
