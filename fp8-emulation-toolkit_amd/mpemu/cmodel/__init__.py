"""mpemu.cmodel -- hardware c-models; `simple` is the one the reference ships (GEMM + direct conv)."""
from . import simple  # noqa: F401
