"""Tuning aid: per-tile cycle stamps of the wave-specialised small-tile GEMM (needs -DFPEMU_WS_TRACE, FPEMU_LIB)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fp8-emulation-toolkit_amd"))
from mpemu.pytquant.hip import _binding as hip
hip._SO = os.environ.get("FPEMU_LIB") or hip._SO
N, C, H, W, K, R, S, st, pd = 32, 64, 56, 56, 64, 3, 3, 1, 1
x = torch.randn(N, C, H, W, device="cuda"); go = torch.randn(N, K, H, W, device="cuda"); gw = torch.empty(K, C, R, S, device="cuda")
L = hip.lib(); L.fpemu_test_ws_trace_.argtypes = [ctypes.c_void_p]; L.fpemu_test_ws_trace_.restype = ctypes.c_int
for _ in range(2):
    hip.conv2d_wu_(gw, go, x, st, pd)
torch.cuda.synchronize()
tr = torch.zeros(256 * 8, dtype=torch.int64, device="cuda")
assert L.fpemu_test_ws_trace_(tr.data_ptr()) == 0
hip.conv2d_wu_(gw, go, x, st, pd)
torch.cuda.synchronize(); L.fpemu_test_ws_trace_(None)
t = tr.cpu().numpy().reshape(256, 8).astype(np.float64)
t0 = t[8, 0]
print("tile: chain[q0 start, before barrier, after barrier]  loader[step start, stores issued, fetch issued, after barrier]  (cycles of s_memtime, relative)")
for it in range(8, 20):
    print(it, [int(t[it, k] - t0) for k in (0, 1, 2)], [int(t[it, k] - t0) for k in (4, 5, 6, 7)])
d = np.diff(t[8:200, 0]); print("chain tile period: median %.0f" % np.median(d))
print("chain: q0..barrier arrival median %.0f, wait at barrier %.0f" % (np.median(t[8:200, 1] - t[8:200, 0]), np.median(t[8:200, 2] - t[8:200, 1])))
print("loader: store issue %.0f, fetch issue %.0f, wait at barrier %.0f" % (np.median(t[8:200, 5] - t[8:200, 4]), np.median(t[8:200, 6] - t[8:200, 5]), np.median(t[8:200, 7] - t[8:200, 6])))
