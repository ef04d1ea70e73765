"""mpemu.pytquant -- operator modules of the fpemu cast (MI355X build).

Keeps the reference's import surface (mpemu/pytquant/__init__.py:4-13):
`from mpemu.pytquant import fpemu_cuda, fpemu_cpp`, each a module with an `FPEmuOp`.
`fpemu_cuda` is the gfx950 HIP operator; `fpemu_cpp` only raises (no CPU path in this build).
"""
from . import cpp
from .cpp import fpemu_cpp

from . import hip
from .hip import fpemu_hip as fpemu_cuda
from .hip import fpemu_hip
