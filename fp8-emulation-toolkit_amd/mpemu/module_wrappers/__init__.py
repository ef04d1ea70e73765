"""mpemu.module_wrappers -- nn.Module shells around functional ops so that quantisation hooks can
attach to them (reference mpemu/module_wrappers/{matmul,eltwise,aggregate,adasparse}.py).  No
kernels live here: the shells call the stock torch op; their inputs/outputs are cast by the hooks."""
from .ops import (AddMatmul, BatchMatmul, EltwiseAdd, EltwiseDiv, EltwiseMul, Matmul, Mean, Norm,
                  SparseConv2d, SparseLinear)
