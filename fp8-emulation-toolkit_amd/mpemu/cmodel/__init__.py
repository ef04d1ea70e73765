from . import simple
