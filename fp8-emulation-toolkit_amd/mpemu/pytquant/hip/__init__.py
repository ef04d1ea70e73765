from . import fpemu as fpemu_hip
