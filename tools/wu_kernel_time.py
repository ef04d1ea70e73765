"""Tuning aid: the weight-update GEMM alone (the chain kernel behind conv2d_wu) timed by HIP events; FPEMU_LIB selects a build."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fp8-emulation-toolkit_amd"))
from mpemu.pytquant.hip import _binding as hip
hip._SO = os.environ.get("FPEMU_LIB") or hip._SO
L = hip.lib()
SHAPES = ((64, 576, 3136, 32), (128, 1152, 784, 32), (256, 2304, 196, 32), (64, 147, 12544, 32))
if os.environ.get('WU_SHAPES'):
    SHAPES = tuple(tuple(int(v) for v in sh.split('x')) for sh in os.environ['WU_SHAPES'].split(','))
for (K, crs, P, nimg) in SHAPES:
    go = torch.randn(nimg, K, P, device="cuda")
    cols = torch.randn(nimg, crs, P, device="cuda")
    G = torch.zeros(K, crs, device="cuda")
    RES = int(os.environ.get("WU_RESIDENT", "0"))   # 1: every k segment re-reads image 0's B (L2-resident); 2: A as well
    def run():
        return L.fpemu_gemm_chain_f32(G.data_ptr(), go.data_ptr(), cols.data_ptr(), K, crs, P, P, P, crs, 0, 1, 1, 1, 0, 0, 0,
                                      nimg, 0 if RES >= 2 else K * P, 0 if RES >= 1 else crs * P, None)
    for _ in range(int(os.environ.get('WU_WARM', '3'))):
        assert run() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    IT = int(os.environ.get('WU_ITERS', '10'))
    for _ in range(IT):
        run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / IT
    steps = nimg * ((P + 127) // 128) * 32
    print("%s  %dx%d chains of %d k: %.3f ms  %.1f ns per 16x16x4 step" % (os.path.basename(hip._SO), K, crs, nimg * P, ms, ms * 1e6 / steps), flush=True)
