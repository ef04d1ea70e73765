"""mpemu -- MI355X-native build of the FP8-Emulation-Toolkit's fpemu cast path.

Import surface of the reference package (mpemu/__init__.py) for the cast path:
    mpemu.pytquant.{fpemu_cuda, fpemu_cpp}.FPEmuOp, mpemu.qutils.*, mpemu.module_wrappers.*,
    mpemu.scale_shift, mpemu.{e5m2,e4m3,e3m4,hybrid}_emu, mpemu.mpt_emu, mpemu.cmodel.simple
The arithmetic runs in hand-written gfx950 HIP kernels (libfpemu_hip.so, C ABI in
include/fpemu_hip.h); see DESIGN.md.  New here: `mpemu.sharded` (multi-GPU shards) and `mpemu.graphs`
(a hooked forward replayed as one HIP graph).
"""
from . import pytquant
from . import cmodel
from . import module_wrappers
from . import qutils
from . import scale_shift
from . import sharded
from . import graphs
from . import mpt_emu
