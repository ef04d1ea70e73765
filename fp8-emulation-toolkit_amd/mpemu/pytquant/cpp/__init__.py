from . import fpemu as fpemu_cpp
