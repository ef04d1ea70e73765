"""Import-compatibility shim: the reference exposes `mpemu.pytquant.cpp.fpemu_cpp`; this build has no CPU
path (FPEmuOp here raises on CPU tensors)."""
from . import fpemu as fpemu_cpp  # noqa: F401
