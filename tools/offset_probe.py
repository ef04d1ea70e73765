"""Does the placement of the output relative to the input move the 1 GiB cast?  One allocation holds both; the output starts
`off` bytes past the input's end.  Interleaved rounds, median of 7 x 20 launches per offset.
usage: offset_probe.py [MiB]"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fp8-emulation-toolkit_amd"))
from mpemu.pytquant.hip import _binding as hip  # noqa: E402

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n = mib * (1 << 20) // 4
OFFS = [0, 256, 1024, 4096, 16384, 65536, 1 << 18, 1 << 20, 3 << 20, 17 << 20, (64 << 20) + 4096 + 256]
slack = max(OFFS) // 4 + 64
buf = torch.empty(2 * n + slack, device="cuda")
buf[:n].normal_()
x = buf[:n]
f, r = hip.parse_mode("E4M3_RNE")


def timed(y, reps=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        hip.cast_(x, y, f, r)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


outs = {off: buf[n + off // 4: 2 * n + off // 4] for off in OFFS}
outs["in-place"] = x
other = torch.empty(n, device="cuda")
outs["separate allocation (%+d B)" % (other.data_ptr() - x.data_ptr() - 4 * n)] = other
for y in outs.values():
    for _ in range(8):
        hip.cast_(x, y, f, r)
torch.cuda.synchronize()
ms = {k: [] for k in outs}
for _ in range(7):
    for k, y in outs.items():
        if k == "in-place":
            continue
        ms[k].append(timed(y))
ms["in-place"] = [timed(x) for _ in range(7)]   # last: it overwrites the input
for k, v in ms.items():
    m = statistics.median(v)
    print("out = in_end + %-34s median %.4f ms  %7.1f GB/s   best %.4f" % (k, m, 8.0 * n / m / 1e6, min(v)))
