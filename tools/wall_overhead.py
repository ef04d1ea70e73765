"""What the wall clock around K back-to-back 1 GiB casts holds besides the K kernels: the same timed region as
bench.py's (sync | events + K launches | sync) in variants, each repeated; prints wall and event ms per step."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "fp8-emulation-toolkit_amd"))
from mpemu.pytquant.hip import _binding as hip  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << 28
x = torch.randn(n, device="cuda")
y = torch.empty_like(x)
f, r = hip.parse_mode("E4M3_RNE")


def step():
    hip.cast_(x, y, f, r, scale=1.0, seed=1, offset=0, elem_base=0)


for _ in range(80):
    step()
torch.cuda.synchronize()


def region(variant):
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    if variant == "t0_after_ev0":
        ev0.record()
        t0 = time.perf_counter()
    else:
        t0 = time.perf_counter()
        if variant != "no_events":
            ev0.record()
    for _ in range(K):
        step()
    if variant != "no_events":
        ev1.record()
    if variant in ("spin", "t0_after_ev0"):
        while not ev1.query():
            pass
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    wall = (t1 - t0) * 1e3 / K
    evt = ev0.elapsed_time(ev1) / K if variant != "no_events" else float("nan")
    return wall, evt


EV = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
EV[0].record()
EV[1].record()


def region_pre():
    """events created (and recorded once) before the region"""
    ev0, ev1 = EV
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(K):
        step()
    ev1.record()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    return (t1 - t0) * 1e3 / K, ev0.elapsed_time(ev1) / K


res = [region_pre() for _ in range(15)]
w = sorted(v[0] for v in res)
e = sorted(v[1] for v in res)
print("%-14s wall ms/step median %.5f min %.5f | events median %.5f | wall - events = %.1f us per region" % (
    "pre-created", w[7], w[0], e[7], (w[7] - e[7]) * K * 1e3), flush=True)
for variant in ("current", "no_events", "spin", "t0_after_ev0", "current"):
    res = [region(variant) for _ in range(15)]
    w = sorted(v[0] for v in res)
    e = sorted(v[1] for v in res)
    print("%-14s wall ms/step median %.5f min %.5f | events median %.5f | wall - events = %.1f us per region" % (
        variant, w[7], w[0], e[7], (w[7] - e[7]) * K * 1e3), flush=True)
