"""bench.py -- throughput of the fpemu cast hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over one batch of synthetic input: the out-of-place
E4M3_RNE cast (scale 1.0) of a 1 GiB fp32 tensor (BASELINE.json configs[1]) resident in HBM.
With N > 1 every rank casts its own 1 GiB shard of an N GiB tensor (weak scaling); unscaled
casts shard with no collective.  Rank 0 prints ONE JSON line.

Extra (untimed for `value`) sections in the same line: per-mode rates for every format x
rounding mode, the per-tensor-scaled path (amax + all-reduce(max) over RCCL when N > 1 + cast),
the `roofline` of the dominant kernel from HIP events over the timed region, and `cpu_baseline`: the
reference's own fpemu_cpp (oracle/_ref, prebuilt from /root/reference; kind "reference") timed on this box's
host cores on a sample of the same data, with the C restatement (kind "port", OpenMP) beside it -- the port
alone where the prebuilt extension is absent or does not run on this host.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (os.path.join(ROOT, "fp8-emulation-toolkit_amd"), os.path.join(ROOT, "oracle")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
SPINUP_STEPS = 64      # untimed launches before the warmup steps (about 20 ms): steady-state clocks
BYTES_PER_ELEM = 8     # fp32 read + fp32 write, unscaled or given-scale cast


def synth(n, device, seed):
    """BASELINE config-2 mix, generated on device: 90 % log-normal magnitudes, 5 % exact
    zeros / format maxima / ties, 5 % raw random bit patterns (NaN / Inf / denormals)."""
    g = torch.Generator(device=device).manual_seed(seed)
    x = torch.randn(n, device=device, generator=g)
    x.mul_(torch.exp(torch.randn(n, device=device, generator=g) * 4.0))
    sel = torch.rand(n, device=device, generator=g)
    special = torch.tensor([0.0, -0.0, 448.0, -448.0, 57344.0, -57344.0, 30.0, -30.0, 65504.0, 1.0625,
                            1.125, 1.25, 2.0 ** -9, 2.0 ** -10, 2.0 ** -16, 464.0], device=device)
    pick = torch.randint(0, special.numel(), (n,), device=device, generator=g)
    x = torch.where(sel < 0.05, special[pick], x)
    del pick
    raw = torch.randint(-2 ** 31, 2 ** 31 - 1, (n,), device=device, generator=g, dtype=torch.int64)
    x = torch.where(sel > 0.95, raw.to(torch.int32).view(torch.float32), x)
    return x.contiguous()


def time_launches(fn, steps):
    """Average device time of one launch from HIP events recorded on the launch stream."""
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    ev[0].record()
    for i in range(steps):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    per = [ev[i].elapsed_time(ev[i + 1]) for i in range(steps)]
    return sum(per) / len(per), min(per)


def reference_baseline(sample, mode, gpu_head):
    """The reference's own fpemu_cpp (oracle/_ref/native, built from /root/reference in the build container)
    timed on this host in a child process; None when the prebuilt extension is absent or does not run here."""
    import subprocess
    import tempfile

    import numpy as np

    so = os.path.join(ROOT, "oracle", "_ref", "native", "fpemu_ref_native.so")
    if not os.path.isfile(so):
        return None
    with tempfile.TemporaryDirectory(prefix="fpemu_bench_") as tmp:
        src, dst = os.path.join(tmp, "in.npy"), os.path.join(tmp, "head.npy")
        np.save(src, sample)
        try:
            r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "ref_time.py"), src, dst, mode, "12"],
                               capture_output=True, text=True, timeout=240)
            if r.returncode != 0:
                return None
            res = json.loads(r.stdout.strip().splitlines()[-1])
            head = np.load(dst).view(np.uint32)
        except Exception:
            return None
    oop, inp = res["out_of_place"], res["in_place"]
    return {"value": round(oop["elements_per_s"], 1), "unit": "elements/s",
            "cores": 1 if mode.startswith(("E4M3", "E3M4")) else res["torch_threads"], "kind": "reference",
            "sample": "%s over the first %d elements of the same tensor through the reference's fpemu_cpp.forward, "
                      "out-of-place as its FPEmuOp calls it (zeros_like + cast): %d passes in %.2f s; its E4M3/E3M4 "
                      "AVX-512 routine runs on one thread (no OpenMP pragma, cpp/fpemu_impl.cpp:934), torch threads "
                      "for the allocation: %d" % (mode, res["elements"], oop["passes"], oop["seconds"],
                                                  res["torch_threads"]),
            "in_place_value": round(inp["elements_per_s"], 1),
            "gpu_matches_reference_on_first_4Mi": bool(np.array_equal(gpu_head, head[: gpu_head.size]))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--elems", type=int, default=1 << 28, help="fp32 elements per GPU (default 1 GiB)")
    ap.add_argument("--mode", default="E4M3_RNE")
    ap.add_argument("--no-extras", action="store_true", help="skip per-mode table and cpu baseline")
    args = ap.parse_args()

    if os.environ.get("FPEMU_BENCH_WATCHDOG"):  # debugging aid: dump every thread's stack and exit after N seconds
        import faulthandler

        faulthandler.dump_traceback_later(float(os.environ["FPEMU_BENCH_WATCHDOG"]), exit=True)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # (FPEMU_BENCH_BACKEND=gloo with all ranks on one GPU rehearses the multi-process flow on a 1-GPU box)
    backend = os.environ.get("FPEMU_BENCH_BACKEND", "nccl")
    local = local % max(torch.cuda.device_count(), 1) if backend != "nccl" else local
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from mpemu.pytquant.hip import _binding as hip

    hip.lib()
    n = args.elems
    x = synth(n, dev, 1234 + rank)
    y = torch.empty_like(x)
    fmt, rm = hip.parse_mode(args.mode)

    def step():
        hip.cast_(x, y, fmt, rm, scale=1.0, seed=1, offset=0, elem_base=rank * n)

    def barrier():
        if world > 1:
            dist.barrier()

    def all_reduce_max(t):
        """RCCL reduces the device tensor in stream order.  The gloo rehearsal (ranks sharing one GPU) stages
        through the host: gloo's own CUDA path hangs on back-to-back calls in that set-up."""
        if backend == "nccl":
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        else:
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.MAX)
            t.copy_(h)

    # Clock spin-up, untimed, ahead of the W warmup steps: the first ~40 launches after an idle GPU run 2-5 %
    # slower (tools/rampup.py: launches 0-19 include the cold launch, 20-39 6.59 TB/s, from 40 on 6.72-6.74),
    # and W is the driver's to choose.
    for _ in range(SPINUP_STEPS):
        step()
    for _ in range(args.warmup):
        step()
    barrier()
    torch.cuda.synchronize()
    # The timed region: EXACTLY `steps` launches.  One HIP event after each launch, recorded on the stream
    # the kernels run on (torch's current stream, which the binding passes to the C ABI): the wall clock gives
    # `value`, the events give the dominant kernel's launch duration over the same launches (`roofline`).
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(args.steps):
        step()
        ev[i + 1].record()
    torch.cuda.synchronize()
    barrier()
    t1 = time.perf_counter()
    ms = (t1 - t0) * 1e3 / args.steps
    if world > 1:
        t = torch.tensor([ms], device=dev, dtype=torch.float64)
        all_reduce_max(t)
        ms = float(t.item())
    value = world * n / (ms * 1e-3)

    # roofline of the dominant kernel from the events of the timed region
    per = [ev[i].elapsed_time(ev[i + 1]) for i in range(args.steps)]
    avg_ms, min_ms = sum(per) / len(per), min(per)
    achieved = BYTES_PER_ELEM * n / (avg_ms * 1e-3) / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.isfile(tpath) and n == (1 << 28):  # the PMC passes were run on the default workload
        try:
            traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "kernel": "cast_kernel<E4M3,RNE,f32>", "avg_launch_ms": round(avg_ms, 5),
                "min_launch_ms": round(min_ms, 5), "algorithmic_bytes_per_launch": BYTES_PER_ELEM * n,
                "timing": "HIP events on the launch stream, one per launch of the timed region"}

    extras = {}
    cpu_baseline = None
    if not args.no_extras:
        # scaled path: amax (+ all-reduce(max) of one word over RCCL) + cast, 12 B/elem algorithmic
        ws = torch.zeros(1, dtype=torch.int32, device=dev)

        def scaled_step():
            ws.zero_()
            hip.amax_(x, ws)
            if world > 1:
                all_reduce_max(ws)
            hip.cast_scaled_(x, y, fmt, rm, ws, 448.0, clamp=True, seed=1, offset=0, elem_base=rank * n)

        for _ in range(3):
            scaled_step()
        s_avg, _ = time_launches(scaled_step, 10)
        extras["scaled_per_tensor"] = {"ms_per_step": round(s_avg, 5),
                                       "GB/s_per_gpu": round(12 * n / (s_avg * 1e-3) / 1e9, 1),
                                       "collective": "all_reduce(max) of 4 B" if world > 1 else "none"}

        if world == 1:
            # the local fused form (no collective): per-block maxima + cast, two launches
            def fused_step():
                hip.quantize_per_tensor_(x, y, fmt, rm, 448.0)

            for _ in range(3):
                fused_step()
            f_avg, _ = time_launches(fused_step, 10)
            extras["fused_per_tensor"] = {"ms_per_step": round(f_avg, 5),
                                          "GB/s_per_gpu": round(12 * n / (f_avg * 1e-3) / 1e9, 1)}

        # BASELINE configs[4], one rank's part: an 8 GiB fp32 shard (2^31 elements, past 32-bit indexing) of a
        # world x 8 GiB tensor -- local amax, all-reduce(max) of the one word over RCCL, cast with the global scale
        try:
            big = 1 << 31
            xb = torch.empty(big, device=dev)
            for c in range(0, big, 1 << 28):
                xb[c:c + (1 << 28)].normal_()
            yb = torch.empty_like(xb)

            def shard_step():
                ws.zero_()
                hip.amax_(xb, ws)
                if world > 1:
                    all_reduce_max(ws)
                hip.cast_scaled_(xb, yb, fmt, rm, ws, 448.0, clamp=True, seed=1, offset=0, elem_base=rank * big)

            shard_step()
            b_avg, _ = time_launches(shard_step, 3)
            extras["shard_8GiB_scaled"] = {"ms_per_step": round(b_avg, 4),
                                           "GB/s_per_gpu": round(12 * big / (b_avg * 1e-3) / 1e9, 1),
                                           "tensor_GiB": 8 * world,
                                           "collective": "all_reduce(max) of 4 B" if world > 1 else "none"}
            del xb, yb
        except torch.cuda.OutOfMemoryError:
            extras["shard_8GiB_scaled"] = "skipped: out of device memory"
        torch.cuda.empty_cache()

        def inplace_step():
            hip.cast_(y, y, fmt, rm, scale=1.0)

        inplace_step()
        i_avg, _ = time_launches(inplace_step, 10)
        extras["inplace"] = {"ms": round(i_avg, 5), "GB/s": round(8 * n / (i_avg * 1e-3) / 1e9, 1)}

        if rank == 0:
            modes = {}
            for m in ("E5M2_RNE", "E5M2_RTZ", "E5M2_STOCHASTIC", "E4M3_RNE", "E4M3_RTZ", "E4M3_STOCHASTIC",
                      "E3M4_RNE", "E3M4_RTZ", "E3M4_STOCHASTIC", "BFLOAT16_RNE", "BFLOAT16_RTZ",
                      "BFLOAT16_STOCHASTIC"):
                f, r = hip.parse_mode(m)

                def mstep(f=f, r=r):
                    hip.cast_(x, y, f, r, scale=1.0, seed=1, offset=3)

                mstep()
                a, _ = time_launches(mstep, 10)
                modes[m] = round(8 * n / (a * 1e-3) / 1e9, 1)
            extras["modes_GB/s"] = modes

        if rank == 0 and world == 1:
            # per-channel rows kernel: 8 B/elem algorithmic (rows of <= 8192 fp32 elements are held
            # in registers between the row maximum and the cast: one read + one write)
            xr = x[: 1 << 26]
            yr = y[: 1 << 26]

            def rows_step():
                hip.quantize_rows_(xr, yr, 16384, fmt, rm, 448.0)

            rows_step()
            r_avg, _ = time_launches(rows_step, 10)
            extras["rows_16384x4096"] = {"ms": round(r_avg, 5), "GB/s": round(8 * xr.numel() / (r_avg * 1e-3) / 1e9, 1)}

            # 16-bit storage (4 B/elem algorithmic) and block norm (8 B/elem: the block is held in registers
            # between its maximum and its cast, one read + one write)
            xb = x[: 1 << 27].to(torch.bfloat16)
            yb = torch.empty_like(xb)

            def bf16_step():
                hip.cast_(xb, yb, fmt, rm, scale=1.0)

            bf16_step()
            b_avg, _ = time_launches(bf16_step, 10)
            extras["bf16_storage_128Mi"] = {"ms": round(b_avg, 5), "GB/s": round(4 * xb.numel() / (b_avg * 1e-3) / 1e9, 1)}
            del xb, yb

            def bn_step():
                hip.quantize_blocks_(xr, yr, 64, fmt, rm)

            bn_step()
            n_avg, _ = time_launches(bn_step, 10)
            extras["blocknorm_64_64Mi"] = {"ms": round(n_avg, 5), "GB/s": round(8 * xr.numel() / (n_avg * 1e-3) / 1e9, 1)}

            # cmodel emulated GEMM, BASELINE config 5's 8192^3: MFMA fmaf chain and VALU
            # separate-rounding chain, each against its own roofline (fp32 MFMA 157.3 / VALU mul+add 78.6)
            M = 8192
            ga = torch.randn(M, M, device=dev)
            gb = torch.randn(M, M, device=dev)
            gc = torch.zeros(M, M, device=dev)
            for name, variant, peak in (("gemm_8192_mfma_fma_chain", 1, 157.3), ("gemm_8192_valu_separate", 0, 78.6)):
                def gstep(variant=variant):
                    hip.gemm_(gc, ga, gb, 1.0, False, False, variant)

                gstep()
                g_avg, _ = time_launches(gstep, 5)
                tf = 2.0 * M ** 3 / (g_avg * 1e-3) / 1e12
                extras[name] = {"ms": round(g_avg, 4), "TFLOP/s": round(tf, 2), "peak": peak,
                                "frac": round(tf / peak, 4)}
            del ga, gb, gc

    if rank == 0 and world == 1 and not args.no_extras:
        import numpy as np

        import orc

        sample = min(n, 1 << 28)
        hx = x[:sample].cpu().numpy()
        orc.cast(hx[: 1 << 24], args.mode)  # warm pass (threads, page tables)
        out = np.empty_like(hx)
        out.fill(0)
        passes, cpu_s = 0, 0.0
        while passes < 4 or (cpu_s * orc.num_threads() < 15.0 and passes < 40):  # ~15+ core-seconds of work
            c0 = time.perf_counter()
            orc.cast(hx, args.mode, out=out)
            cpu_s += time.perf_counter() - c0
            passes += 1
        step()
        torch.cuda.synchronize()
        chk = slice(0, 1 << 22)
        gpu_head = y[chk].cpu().numpy().view(np.uint32)
        same = bool(np.array_equal(gpu_head, out[chk].view(np.uint32)))
        port = {"value": round(sample * passes / cpu_s, 1), "unit": "elements/s",
                "cores": orc.num_threads(), "kind": "port",
                "sample": "%s over the first %d elements of the same tensor, %d timed passes (%.2f s wall) of "
                          "the OpenMP C oracle (oracle/fpemu_oracle.c), output pre-faulted" % (
                              args.mode, sample, passes, cpu_s),
                "gpu_matches_oracle_on_first_4Mi": same}
        cpu_baseline = reference_baseline(hx[: 1 << 26], args.mode, gpu_head) or port
        if cpu_baseline is not port:
            cpu_baseline["port"] = port

    if rank == 0:
        line = {
            "metric": "quantized elements/sec, %s fpemu cast of a 1 GiB fp32 tensor per GPU" % args.mode,
            "value": value, "unit": "elements/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "spinup_steps": SPINUP_STEPS, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: %s out-of-place cast, %d fp32 elements (%.2f GiB) "
                                   "per GPU, scale 1.0, inputs resident in HBM" % (args.mode, n, n * 4 / 2 ** 30),
                       "sharding": "dp%d: contiguous shards, no collective on the unscaled path" % world},
            "hbm_GB/s_per_gpu": round(BYTES_PER_ELEM * n / (ms * 1e-3) / 1e9, 1),
            "hbm_frac_of_peak": round(BYTES_PER_ELEM * n / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "roofline": roofline, "cpu_baseline": cpu_baseline, "extras": extras,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()  # rank 0's extras run longer than the others': leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
