"""bench.py -- throughput of the fpemu cast hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over one batch of synthetic input: the out-of-place
E4M3_RNE cast (scale 1.0) of a 1 GiB fp32 tensor (BASELINE.json configs[1]) resident in HBM.
With N > 1 every rank casts its own 1 GiB shard of an N GiB tensor (weak scaling); unscaled
casts shard with no collective.  Rank 0 prints ONE JSON line.  `--gpus N` without a
launcher (WORLD_SIZE unset) starts the N ranks itself through torch.distributed.run; under a launcher
the world size must equal N.

Extra (untimed for `value`) sections in the same line, every one of them CHECKED against the oracle on
a head of its output (`verified`): per-mode rates for every format x rounding mode, the per-tensor-scaled
path (amax + all-reduce(max) over RCCL when N > 1 + cast), rows / block norm / 16-bit storage, the c-model
GEMM and convolution, `ceilings` (bare read / write / copy streams timed in this process), the `roofline`
of the dominant kernel from HIP events over the timed region, and `cpu_baseline`: the reference's own
fpemu_cpp (oracle/_ref, prebuilt from /root/reference; kind "reference") timed on this box's host cores
on a sample of the same data, with the C restatement (kind "port", OpenMP) beside it.  With N > 1 rank 0
also re-casts every rank's shard on its own GPU and compares 64-bit checksums (`extras.sharded_checksums`):
the N-GPU result is the 1-GPU result of the same global tensor.
"""
import argparse
import ctypes
import hashlib
import json
import os
import socket
import subprocess
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
HEAD = 1 << 22         # elements of every extras leg's output compared with the oracle

CONFIG2_MODES = ("E5M2_RNE", "E5M2_RTZ", "E5M2_STOCHASTIC", "E4M3_RNE", "E4M3_RTZ", "E4M3_STOCHASTIC",
                 "E3M4_RNE", "E3M4_RTZ", "E3M4_STOCHASTIC", "BFLOAT16_RNE", "BFLOAT16_RTZ", "BFLOAT16_STOCHASTIC")


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


def time_batch(fn, launches=20, warm=3):
    """Average device time of one launch from TWO HIP events around `launches` back-to-back launches (an event per
    launch puts a marker packet, 5-10 us on this runtime, inside every interval: visible for 50-100 us kernels)."""
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(launches):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / launches


def checksum64(t):
    """Two 64-bit sums over the bit patterns of a tensor (plain and position-weighted), wrapping."""
    b = t.reshape(-1).view(torch.int32 if t.element_size() == 4 else torch.int16)
    s1 = torch.zeros((), dtype=torch.int64, device=t.device)
    s2 = torch.zeros((), dtype=torch.int64, device=t.device)
    step = 1 << 26
    w = (torch.arange(step, device=t.device, dtype=torch.int64) % 8191) + 1
    for c in range(0, b.numel(), step):
        v = b[c:c + step].to(torch.int64)
        s1 += v.sum()
        s2 += (v * w[: v.numel()]).sum()
    return [int(s1.item()), int(s2.item())]


def reference_baseline(sample, mode, gpu_head):
    """The reference's own fpemu_cpp (oracle/_ref/native, built from /root/reference in the build container)
    timed on this host in a child process; None when the prebuilt extension is absent or does not run here."""
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


def launch_ranks(args):
    """`python bench.py --gpus N` with no launcher: start the N ranks (one process per GPU) BEFORE this process
    touches the GPU, and leave with their exit code."""
    backend = os.environ.get("FPEMU_BENCH_BACKEND", "nccl")
    have = torch.cuda.device_count()  # does not initialise the GPU
    if backend == "nccl" and have < args.gpus:
        sys.stderr.write("bench.py: --gpus %d but %d GPU(s) visible; one rank per GPU over RCCL needs %d.\n"
                         "(rehearse the multi-process flow on one GPU with FPEMU_BENCH_BACKEND=gloo)\n"
                         % (args.gpus, have, args.gpus))
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def load_ceilings():
    so = os.path.join(ROOT, "tools", "libfpemu_ceilings.so")
    if not os.path.isfile(so):
        return None
    L = ctypes.CDLL(so)
    vp, u64 = ctypes.c_void_p, ctypes.c_uint64
    L.ceil_read.argtypes = [vp, vp, u64, vp]
    L.ceil_write.argtypes = [vp, u64, vp]
    L.ceil_copy.argtypes = [vp, vp, u64, ctypes.c_int, vp]
    L.ceil_mfma.argtypes = [vp, ctypes.c_int, ctypes.c_int, vp]
    L.ceil_valu.argtypes = [vp, ctypes.c_int, ctypes.c_int, vp]
    return L


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--elems", type=int, default=1 << 28, help="fp32 elements per GPU (default 1 GiB)")
    ap.add_argument("--mode", default="E4M3_RNE")
    ap.add_argument("--no-extras", action="store_true", help="skip per-mode table and cpu baseline")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    if os.environ.get("FPEMU_BENCH_WATCHDOG"):  # debugging aid: dump every thread's stack and exit after N seconds
        import faulthandler

        faulthandler.dump_traceback_later(float(os.environ["FPEMU_BENCH_WATCHDOG"]), exit=True)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.stderr.write("bench.py: launched with WORLD_SIZE=%d but --gpus %d: they must agree (python -m "
                         "torch.distributed.run --nnodes=1 --nproc-per-node %d ... bench.py --gpus %d)\n"
                         % (world, args.gpus, args.gpus, args.gpus))
        sys.exit(2)
    # (FPEMU_BENCH_BACKEND=gloo with all ranks on one GPU rehearses the multi-process flow on a 1-GPU box)
    backend = os.environ.get("FPEMU_BENCH_BACKEND", "nccl")
    local = local % max(torch.cuda.device_count(), 1) if backend != "nccl" else local
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # FPEMU_BENCH_REHEARSE=1 (under a launcher, --gpus 1): every collective of the N > 1 flow -- barriers, the timing
    # reduction, the amax all-reduce, the checksum gather -- through a ONE-rank communicator of the chosen backend: the
    # RCCL calls of the multi-GPU bench executed on a box with a single GPU (tools/gpu.sh check)
    multi = world > 1 or (os.environ.get("FPEMU_BENCH_REHEARSE") == "1" and "WORLD_SIZE" in os.environ)
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from mpemu.pytquant.hip import _binding as hip
    from mpemu import sharded
    from mpemu.sharded import all_reduce_max_words

    if multi and world == 1:
        sharded._SINGLE_RANK_SHORTCUT = False
    hip.lib()
    n = args.elems
    # Input and output are the two halves of ONE allocation (the output starts where the input ends): where the
    # allocator happens to put two separate 1 GiB blocks moves the cast by about +-0.7 % (tools/offset_probe.py,
    # profiles/r03_offset_probe.txt: 6.80-6.90 TB/s over eleven placements on one box), so the placement is fixed here
    # instead of left to chance.  Both buffers stay caller-owned and distinct: the out-of-place workload as named.
    xy = torch.empty(2 * n, dtype=torch.float32, device=dev)
    x, y = xy[:n], xy[n:]
    x.copy_(synth(n, dev, 1234 + rank))
    fmt, rm = hip.parse_mode(args.mode)
    stream = torch.cuda.current_stream(dev).cuda_stream

    def step():
        hip.cast_(x, y, fmt, rm, scale=1.0, seed=1, offset=0, elem_base=rank * n)

    def barrier():
        if multi:
            dist.barrier()

    def reduce_max_f64(v):
        """MAX of one Python float over the ranks (RCCL on the device; other backends on the host)."""
        if not multi:
            return v
        t = torch.tensor([v], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def gather_i64(vals):
        """Every rank's list of 64-bit integers, on every rank: [world][len(vals)]."""
        if not multi:
            return [list(vals)]
        t = torch.tensor(vals, dtype=torch.int64, device=dev if backend == "nccl" else "cpu")
        outs = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(outs, t)
        return [o.tolist() for o in outs]

    # Clock spin-up, untimed, ahead of the W warmup steps: the first ~40 launches after an idle GPU run 2-5 %
    # slower (tools/rampup.py: launches 0-19 include the cold launch, 20-39 6.59 TB/s, from 40 on 6.72-6.74),
    # and W is the driver's to choose.
    for _ in range(SPINUP_STEPS):
        step()
    # the timed region's two events exist before it starts (torch creates an event at its first record: a
    # hipEventCreate inside the region otherwise, tools/wall_overhead.py)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    ev1.record()
    for _ in range(args.warmup):
        step()
    barrier()
    torch.cuda.synchronize()
    # The timed region: EXACTLY `steps` launches, back to back.  Two HIP events bracket them on the stream the kernels
    # run on (torch's current stream, which the binding passes to the C ABI): the wall clock gives `value`, the events
    # give the dominant kernel's average launch duration over the same launches (`roofline`).  (An event after every
    # launch, as round 1 had it, puts a marker packet between consecutive kernels: ~1 % of a 0.31 ms launch.)
    t0 = time.perf_counter()
    ev0.record()
    for i in range(args.steps):
        step()
    ev1.record()
    torch.cuda.synchronize()
    t1 = time.perf_counter()  # this rank's K steps are done; the closing barrier follows, the MAX over ranks is below
    barrier()                 # (read after the barrier, an RCCL barrier's own launch + sync -- ~0.1-0.2 ms, rehearsed
    #                            with one rank -- would be charged to the K steps: 3 % of a 20-step region)
    ms = reduce_max_f64((t1 - t0) * 1e3 / args.steps)
    value = world * n / (ms * 1e-3)

    # roofline of the dominant kernel from the events of the timed region; the fastest single launch from a short
    # untimed pass with one event per launch
    avg_ms = ev0.elapsed_time(ev1) / args.steps
    _a, min_ms = time_launches(step, 10)
    achieved = BYTES_PER_ELEM * n / (avg_ms * 1e-3) / 1e9
    # HBM traffic from the committed PMC passes (profiles/pmc_traffic.json) -- reported only when they were
    # collected on THIS build of the library (sha256 of libfpemu_hip.so recorded beside them) and workload
    traffic, traffic_note = None, "no PMC record"
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.isfile(tpath) and n == (1 << 28):
        try:
            rec = json.load(open(tpath))
            so_hash = hashlib.sha256(open(hip._SO, "rb").read()).hexdigest()
            if rec.get("lib_sha256") == so_hash:
                traffic, traffic_note = rec.get("hbm_bytes_per_launch"), "rocprofv3 --pmc passes on this library build"
            else:
                traffic_note = "PMC record is from another build of libfpemu_hip.so: not reported"
        except Exception:
            traffic = None
    roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_note,
                "kernel": "cast_kernel<E4M3,RNE,f32>", "avg_launch_ms": round(avg_ms, 5),
                "min_launch_ms": round(min_ms, 5), "algorithmic_bytes_per_launch": BYTES_PER_ELEM * n,
                "timing": "two HIP events on the launch stream around the timed region's launches; min from 10 more launches, one event each"}

    extras = {}
    cpu_baseline = None
    verified = {}
    if not args.no_extras:
        import numpy as np

        import orc

        def same(gpu_t, want, nan_ok=False):
            g = gpu_t.reshape(-1).view(torch.int32).cpu().numpy().view(np.uint32)
            w = np.ascontiguousarray(want).reshape(-1).view(np.uint32)
            if nan_ok:  # a NaN the arithmetic generated: payload is an ISA convention
                both = ((g & 0x7FFFFFFF) > 0x7F800000) & ((w & 0x7FFFFFFF) > 0x7F800000)
                return bool(np.all((g == w) | both))
            return bool(np.array_equal(g, w))

        hx = x[:HEAD].cpu().numpy()
        base = rank * n
        step()
        verified["timed_cast"] = same(y[:HEAD], orc.cast(hx, args.mode, seed=1, offset=0, elem_base=base))

        # ---- ceilings: what the part gives bare streams over the same buffers, in this process -----------------
        C = load_ceilings() if rank == 0 else None
        if C is not None:
            nvec = n // 4
            legs = {"cast": step,
                    "read_only": lambda: C.ceil_read(x.data_ptr(), y.data_ptr(), nvec, stream),
                    "write_only": lambda: C.ceil_write(y.data_ptr(), nvec, stream),
                    "copy_b128": lambda: C.ceil_copy(x.data_ptr(), y.data_ptr(), nvec, 128, stream),
                    "copy_b256": lambda: C.ceil_copy(x.data_ptr(), y.data_ptr(), nvec, 256, stream)}
            ms_c = {k: [] for k in legs}
            for rnd in range(6):  # interleaved rounds, each leg back to back with itself (steady cache state)
                for k, f in legs.items():
                    for _ in range(3):
                        f()
                    a, _m = time_launches(f, 10)
                    if rnd > 0:
                        ms_c[k].append(a)
            byts = {"cast": 8 * n, "read_only": 4 * n, "write_only": 4 * n, "copy_b128": 8 * n, "copy_b256": 8 * n}
            gbs = {k: round(byts[k] / (sorted(v)[len(v) // 2] * 1e-3) / 1e9, 1) for k, v in ms_c.items()}
            best_copy = max(gbs["copy_b128"], gbs["copy_b256"])
            extras["ceilings"] = {"GB/s": gbs, "cast_over_best_copy": round(gbs["cast"] / best_copy, 4),
                                  "how": "tools/ceilings.hip: one 16-byte vector per lane, no loop, nt loads / sc1 nt "
                                         "stores; median of 5 interleaved rounds x 10 launches, same buffers, same process"}
            step()  # y holds the cast again

        # ---- placement: the same cast into the adjacent output (the timed workload) and into a separately allocated one ----
        if rank == 0:
            y_sep = torch.empty(n, dtype=torch.float32, device=dev)

            def sep_step():
                hip.cast_(x, y_sep, fmt, rm, scale=1.0, seed=1, offset=0, elem_base=rank * n)

            pl = {"adjacent": [], "separate": []}
            for _ in range(3):
                pl["adjacent"].append(time_batch(step, launches=20, warm=3))
                pl["separate"].append(time_batch(sep_step, launches=20, warm=3))
            extras["placement"] = {
                "GB/s": {k: round(8 * n / (sorted(v)[1] * 1e-3) / 1e9, 1) for k, v in pl.items()},
                "separate_offset_bytes": y_sep.data_ptr() - x.data_ptr() - 4 * n,
                "how": "output right behind the input in one allocation (what the timed region uses) against an output "
                       "torch allocated on its own; median of 3 interleaved rounds x 20 launches, two events each "
                       "(tools/offset_probe.py sweeps more placements)"}
            del y_sep

        # ---- scaled path: amax (+ all-reduce(max) of one word over RCCL) + cast, 12 B/elem algorithmic ---------
        xf = torch.where(torch.isfinite(x), x, torch.full_like(x, 0.5))  # a NaN would make the scale NaN
        hxf = xf[:HEAD].cpu().numpy()
        ws = torch.zeros(1, dtype=torch.int32, device=dev)

        def scaled_step():
            ws.zero_()
            hip.amax_(xf, ws)
            if multi:
                all_reduce_max_words(ws)
            hip.cast_scaled_(xf, y, fmt, rm, ws, 448.0, clamp=True, seed=1, offset=0, elem_base=base)

        barrier()
        s_avg = reduce_max_f64(time_batch(scaled_step, launches=20, warm=3))
        gmax = reduce_max_f64(float(xf.abs().max().item()))
        word_ok = int(ws.item()) == int(np.float32(gmax).view(np.uint32))
        sc = orc.scale_from_amax(np.float32(gmax), 448.0)
        verified["scaled_per_tensor"] = word_ok and same(y[:HEAD], orc.cast(hxf, args.mode, scale=sc, seed=1, offset=0,
                                                                            elem_base=base))
        # the collective alone: all-reduce(MAX) of the one amax word, back to back (RCCL: two events on the stream
        # the all-reduce is enqueued on; host-staged backends: wall clock around the synchronising calls)
        collective_us = None
        if multi:
            cw = torch.zeros(1, dtype=torch.int32, device=dev)
            for _ in range(5):
                all_reduce_max_words(cw)
            torch.cuda.synchronize()
            barrier()
            if backend == "nccl":
                collective_us = time_batch(lambda: all_reduce_max_words(cw), launches=50, warm=0) * 1e3
            else:
                c0 = time.perf_counter()
                for _ in range(50):
                    all_reduce_max_words(cw)
                collective_us = (time.perf_counter() - c0) / 50 * 1e6
            collective_us = round(reduce_max_f64(collective_us), 2)
        extras["scaled_per_tensor"] = {"ms_per_step": round(s_avg, 5),
                                       "GB/s_per_gpu": round(12 * n / (s_avg * 1e-3) / 1e9, 1),
                                       "elements/s_whole_job": round(world * n / (s_avg * 1e-3), 1),
                                       "collective": "all_reduce(max) of 4 B" if multi else "none",
                                       "collective_us": collective_us,
                                       "timing": "two HIP events around 20 steps (amax + all-reduce + cast), max over ranks",
                                       "verified": verified["scaled_per_tensor"]}

        if not multi:
            # the local fused form (no collective): per-block maxima + cast, two launches
            def fused_step():
                hip.quantize_per_tensor_(xf, y, fmt, rm, 448.0)

            f_avg = time_batch(fused_step, launches=20, warm=3)
            verified["fused_per_tensor"] = same(y[:HEAD], orc.cast(hxf, args.mode, scale=sc))
            extras["fused_per_tensor"] = {"ms_per_step": round(f_avg, 5),
                                          "GB/s_per_gpu": round(12 * n / (f_avg * 1e-3) / 1e9, 1),
                                          "verified": verified["fused_per_tensor"]}

        # ---- N > 1: the N-GPU result is the 1-GPU result of the same global tensor ----------------
        if multi:
            sums = {}
            step()
            sums["unscaled"] = checksum64(y)
            f5, r5 = hip.parse_mode("E5M2_STOCHASTIC")
            hip.cast_(x, y, f5, r5, scale=1.0, seed=7, offset=3, elem_base=base)
            sums["stochastic"] = checksum64(y)
            scaled_step()
            sums["scaled"] = checksum64(y)
            keys = sorted(sums)
            allsums = gather_i64([v for k in keys for v in sums[k]])
            if rank == 0:
                ok = True
                shards = [synth(n, dev, 1234 + r) if r else x for r in range(world)]
                fins = [torch.where(torch.isfinite(s), s, torch.full_like(s, 0.5)) for s in shards]
                w1 = torch.zeros(1, dtype=torch.int32, device=dev)
                for s in fins:
                    hip.amax_(s, w1)  # one GPU, the whole tensor: the word accumulates over the shards
                for r in range(world):
                    mine = {}
                    hip.cast_(shards[r], y, fmt, rm, scale=1.0, seed=1, offset=0, elem_base=r * n)
                    mine["unscaled"] = checksum64(y)
                    hip.cast_(shards[r], y, f5, r5, scale=1.0, seed=7, offset=3, elem_base=r * n)
                    mine["stochastic"] = checksum64(y)
                    hip.cast_scaled_(fins[r], y, fmt, rm, w1, 448.0, clamp=True, seed=1, offset=0, elem_base=r * n)
                    mine["scaled"] = checksum64(y)
                    ok = ok and [v for k in keys for v in mine[k]] == allsums[r]
                del shards, fins
                verified["sharded_checksums"] = ok
                extras["sharded_checksums"] = {"equal_to_single_gpu": ok, "ranks": world, "legs": keys,
                                               "how": "rank 0 regenerates every rank's shard, casts it on its own GPU "
                                                      "with that shard's elem_base / the whole tensor's amax word, and "
                                                      "compares two 64-bit sums per leg with the all-gathered ones"}
            torch.cuda.empty_cache()

        # ---- BASELINE configs[4], one rank's part: an 8 GiB fp32 shard (2^31 elements, past 32-bit indexing) of a
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
                if multi:
                    all_reduce_max_words(ws)
                hip.cast_scaled_(xb, yb, fmt, rm, ws, 448.0, clamp=True, seed=1, offset=0, elem_base=rank * big)

            shard_step()
            b_avg, _ = time_launches(shard_step, 3)
            bmax = 0.0
            for c in range(0, big, 1 << 28):
                bmax = max(bmax, float(xb[c:c + (1 << 28)].abs().max().item()))
            bmax = reduce_max_f64(bmax)
            sb = orc.scale_from_amax(np.float32(bmax), 448.0)
            tail = slice(big - HEAD, big)  # the far end: indices past 2^31 - 2^22
            verified["shard_8GiB_scaled"] = (int(ws.item()) == int(np.float32(bmax).view(np.uint32))
                                             and same(yb[tail], orc.cast(xb[tail].cpu().numpy(), args.mode, scale=sb)))
            extras["shard_8GiB_scaled"] = {"ms_per_step": round(b_avg, 4),
                                           "GB/s_per_gpu": round(12 * big / (b_avg * 1e-3) / 1e9, 1),
                                           "tensor_GiB": 8 * world,
                                           "collective": "all_reduce(max) of 4 B" if multi else "none",
                                           "verified": verified["shard_8GiB_scaled"]}
            del xb, yb
        except torch.cuda.OutOfMemoryError:
            extras["shard_8GiB_scaled"] = "skipped: out of device memory"
        torch.cuda.empty_cache()

        step()

        def inplace_step():
            hip.cast_(y, y, fmt, rm, scale=1.0)

        i_avg = time_batch(inplace_step, launches=10, warm=1)
        verified["inplace"] = same(y[:HEAD], orc.cast(orc.cast(hx, args.mode), args.mode))
        extras["inplace"] = {"ms": round(i_avg, 5), "GB/s": round(8 * n / (i_avg * 1e-3) / 1e9, 1),
                             "verified": verified["inplace"]}

        if rank == 0:
            modes, modes_ok = {}, {}
            for m in CONFIG2_MODES:
                f, r = hip.parse_mode(m)

                def mstep(f=f, r=r):
                    hip.cast_(x, y, f, r, scale=1.0, seed=1, offset=3)

                # (the better of two batches: each mode starts right after host-side checking, with the GPU idle)
                a = min(time_batch(mstep, launches=10, warm=3), time_batch(mstep, launches=10, warm=0))
                modes[m] = round(8 * n / (a * 1e-3) / 1e9, 1)
                modes_ok[m] = same(y[:HEAD], orc.cast(hx, m, seed=1, offset=3))
            extras["modes_GB/s"] = modes
            extras["modes_verified"] = modes_ok
            verified["modes"] = all(modes_ok.values())

        if rank == 0 and not multi:
            # per-channel rows kernel: 8 B/elem algorithmic (the row is held on chip between its
            # maximum and its cast: one read + one write)
            def rows_oracle(h, cols, clamp=True):
                out = np.empty_like(h)
                for r0 in range(0, h.size, cols):
                    row = h[r0:r0 + cols]
                    out[r0:r0 + cols] = orc.cast(row, args.mode, scale=orc.scale_from_amax(orc.amax(row), 448.0, clamp=clamp))
                return out

            for name, rows, cols in (("rows_16384x4096", 16384, 4096), ("rows_4096x11008", 4096, 11008),
                                     ("rows_1024x65536", 1024, 65536)):
                xr, yr = xf[: rows * cols], y[: rows * cols]

                def rows_step(xr=xr, yr=yr, rows=rows):
                    hip.quantize_rows_(xr, yr, rows, fmt, rm, 448.0)

                r_avg = time_batch(rows_step, launches=20, warm=3)
                hrows = max(1, HEAD // cols)
                verified[name] = same(yr[: hrows * cols], rows_oracle(xf[: hrows * cols].cpu().numpy(), cols))
                extras[name] = {"ms": round(r_avg, 5), "GB/s": round(8 * xr.numel() / (r_avg * 1e-3) / 1e9, 1),
                                "verified": verified[name]}

            # 16-bit storage (4 B/elem algorithmic) and block norm (8 B/elem: the block is held in registers
            # between its maximum and its cast, one read + one write)
            xb = x[: 1 << 27].to(torch.bfloat16)
            yb = torch.empty_like(xb)

            def bf16_step():
                hip.cast_(xb, yb, fmt, rm, scale=1.0)

            b_avg = time_batch(bf16_step, launches=20, warm=3)
            hb = xb[:HEAD].view(torch.int16).cpu().numpy().view(np.uint16)
            gb = yb[:HEAD].view(torch.int16).cpu().numpy().view(np.uint16)
            wb = orc.cast16(hb, True, args.mode)
            isnan = lambda a: (a & 0x7FFF) > 0x7F80
            verified["bf16_storage_128Mi"] = bool(np.all((gb == wb) | (isnan(gb) & isnan(wb))))
            extras["bf16_storage_128Mi"] = {"ms": round(b_avg, 5), "GB/s": round(4 * xb.numel() / (b_avg * 1e-3) / 1e9, 1),
                                            "verified": verified["bf16_storage_128Mi"]}
            del xb, yb

            xr, yr = x[: 1 << 26], y[: 1 << 26]

            def bn_step():
                hip.quantize_blocks_(xr, yr, 64, fmt, rm)

            n_avg = time_batch(bn_step, launches=20, warm=3)
            verified["blocknorm_64_64Mi"] = same(yr[:HEAD], orc.cast_blocknorm(hx, args.mode, 64), nan_ok=True)
            extras["blocknorm_64_64Mi"] = {"ms": round(n_avg, 5), "GB/s": round(8 * xr.numel() / (n_avg * 1e-3) / 1e9, 1),
                                           "verified": verified["blocknorm_64_64Mi"]}

            # cmodel emulated GEMM, BASELINE config 5's 8192^3: MFMA fmaf chain and VALU
            # separate-rounding chain, each against its own roofline (fp32 MFMA 157.3 / VALU mul+add 78.6);
            # a 64 x 64 grid of sampled outputs (full 8192-step chains) against the sequential-k oracle
            M = 8192
            ga = torch.randn(M, M, device=dev)
            gb_ = torch.randn(M, M, device=dev)
            gc = torch.zeros(M, M, device=dev)
            si = torch.arange(5, M, 129, device=dev)[:64]
            sj = torch.arange(3, M, 127, device=dev)[:64]
            ha, hbm = ga[si].cpu().numpy(), gb_[:, sj].contiguous().cpu().numpy()
            # compute ceilings timed in this process, right before the legs they price: the bare MFMA pipe and the
            # bare packed multiply + add stream (tools/ceilings.hip), 10 untimed launches then two events around 10
            cc = {}
            if C is not None:
                sink = torch.zeros(1024, device=dev)
                it_m, it_v, blk_m, blk_v = 2000, 4000, 512, 1024
                cm = time_batch(lambda: C.ceil_mfma(sink.data_ptr(), it_m, blk_m, stream), launches=10, warm=10)
                cv = time_batch(lambda: C.ceil_valu(sink.data_ptr(), it_v, blk_v, stream), launches=10, warm=10)
                cc["mfma_f32"] = round(blk_m * 4 * it_m * 64.0 * 4096.0 / (cm * 1e-3) / 1e12, 2)
                cc["valu_pk_f32"] = round(blk_v * 256.0 * it_v * 4 * 16 * 2 * 2 / (cv * 1e-3) / 1e12, 2)
                extras.setdefault("ceilings", {})["TFLOP/s"] = dict(
                    cc, how="bare v_mfma_f32_32x32x2_f32 chains (4 per wave, 2 waves per SIMD) / bare v_pk_mul_f32 + "
                            "v_pk_add_f32 chains (16 per lane, 4 waves per SIMD), no memory traffic; same process, "
                            "10 untimed launches then two events around 10")
            for name, variant, peak, fused, bare in (("gemm_8192_mfma_fma_chain", 1, 157.3, True, "mfma_f32"),
                                                     ("gemm_8192_valu_separate", 0, 78.6, False, "valu_pk_f32")):
                def gstep(variant=variant):
                    hip.gemm_(gc, ga, gb_, 1.0, False, False, variant)

                g_avg = time_batch(gstep, launches=10, warm=10)
                _a, g_min = time_launches(gstep, 5)
                gc.zero_()
                gstep()
                verified[name] = same(gc[si][:, sj].contiguous(), orc.gemm(ha, hbm, 1.0, fused=fused))
                tf = 2.0 * M ** 3 / (g_avg * 1e-3) / 1e12
                extras[name] = {"ms": round(g_avg, 4), "min_ms": round(g_min, 4), "TFLOP/s": round(tf, 2), "peak": peak,
                                "frac": round(tf / peak, 4),
                                "frac_of_bare_pipe": round(tf / cc[bare], 4) if bare in cc else None,
                                "timing": "10 untimed launches, then two HIP events around 10; min from 5 more, one event each",
                                "verified": verified[name]}
            # per-tensor quantisation of the 256 MiB product (quantize_module_output, reference qutils.py:607-609): amax pass
            # + cast, against the cast alone from the maximum the GEMM's epilogue found (fpemu_gemm_f32_amax)
            wq = torch.zeros(1, dtype=torch.int32, device=dev)
            qf, qr = hip.parse_mode("E4M3_RNE")
            gc.zero_()
            hip.gemm_(gc, ga, gb_, 1.0, False, False, 1, amax=wq)
            gq, gq2 = torch.empty_like(gc), torch.empty_like(gc)
            ws_q = torch.empty(257, dtype=torch.int32, device=dev)
            q_two = time_batch(lambda: hip.quantize_per_tensor_(gc, gq2, qf, qr, 448.0, amax_ws=ws_q), launches=20, warm=3)
            q_one = time_batch(lambda: hip.cast_scaled_(gc, gq, qf, qr, wq, 448.0), launches=20, warm=3)
            verified["gemm_output_quantise"] = bool(torch.equal(gq.view(torch.int32), gq2.view(torch.int32))) and same(
                gq[si][:, sj].contiguous(),
                orc.cast(orc.gemm(ha, hbm, 1.0, fused=True), "E4M3_RNE", scale=orc.scale_from_amax(
                    np.array([int(wq.item())], np.uint32).view(np.float32)[0], 448.0)))
            extras["gemm_output_quantise"] = {
                "amax_pass_plus_cast_ms": round(q_two, 5), "cast_from_epilogue_amax_ms": round(q_one, 5),
                "GB/s": round(8 * gc.numel() / (q_one * 1e-3) / 1e9, 1), "tensor_MiB": gc.numel() * 4 >> 20,
                "verified": verified["gemm_output_quantise"]}
            del ga, gb_, gc, gq, gq2

            # cmodel convolution: ResNet-50's 3x3, 64 -> 64 channels, 56 x 56, batch 32
            try:
                N_, C_, H_, K_ = 32, 64, 56, 64
                ci = torch.randn(N_, C_, H_, H_, device=dev)
                cw = torch.randn(K_, C_, 3, 3, device=dev) * 0.05
                cb = torch.randn(K_, device=dev)
                co = torch.empty(N_, K_, H_, H_, device=dev)

                def conv_step():
                    hip.conv2d_fp_(co, ci, cw, cb, 1, 1)

                conv_step()
                c_avg, _ = time_launches(conv_step, 10)
                want = orc.conv2d_fp(ci[:1].cpu().numpy(), cw.cpu().numpy(), cb.cpu().numpy(), 1, 1, fused=True)
                verified["conv_fp_3x3_c64_56"] = same(co[:1].contiguous(), want)
                fl = 2.0 * N_ * K_ * H_ * H_ * C_ * 9
                extras["conv_fp_3x3_c64_56"] = {"ms": round(c_avg, 5), "TFLOP/s_effective": round(fl / (c_avg * 1e-3) / 1e12, 2),
                                                "verified": verified["conv_fp_3x3_c64_56"]}
                # the weight update of the same layer: a chain over all N * OH * OW products per weight (small-tile chain GEMM)
                go = torch.randn(N_, K_, H_, H_, device=dev)
                gw = torch.empty_like(cw)

                def wu_step():
                    hip.conv2d_wu_(gw, go, ci, 1, 1)

                wu_step()
                w_avg, _ = time_launches(wu_step, 5)
                gw2 = torch.empty_like(cw)
                hip.conv2d_wu_(gw2, go[:2].contiguous(), ci[:2].contiguous(), 1, 1)  # two images: what the oracle checks in seconds
                want = orc.conv2d_wu(go[:2].cpu().numpy(), ci[:2].cpu().numpy(), (3, 3), 1, 1, fused=True)
                verified["conv_wu_3x3_c64_56"] = same(gw2, want)
                extras["conv_wu_3x3_c64_56"] = {"ms": round(w_avg, 5), "TFLOP/s_effective": round(fl / (w_avg * 1e-3) / 1e12, 2),
                                                "verified": verified["conv_wu_3x3_c64_56"], "verified_on": "2 of the 32 images (same kernel)"}
                del ci, cw, cb, co, go, gw, gw2
                # backward-data of a STRIDED layer (3x3 stride 2, 128 -> 128 channels, 56 x 56 -> 28 x 28): one chain GEMM per
                # phase of the stride over the taps that exist
                sw = torch.randn(128, 128, 3, 3, device=dev) * 0.05
                sgo = torch.randn(N_, 128, 28, 28, device=dev)
                sgi = torch.empty(N_, 128, 56, 56, device=dev)

                def bp_step():
                    hip.conv2d_bp_(sgi, sgo, sw, 2, 1)

                bp_step()
                b_avg, _ = time_launches(bp_step, 5)
                want = orc.conv2d_bp(sgo[:1].cpu().numpy(), sw.cpu().numpy(), (56, 56), 2, 1, fused=True)
                verified["conv_bp_3x3_s2_c128_56"] = same(sgi[:1].contiguous(), want)
                fl2 = 2.0 * N_ * 128 * 28 * 28 * 128 * 9
                extras["conv_bp_3x3_s2_c128_56"] = {"ms": round(b_avg, 5), "TFLOP/s_effective": round(fl2 / (b_avg * 1e-3) / 1e12, 2),
                                                    "verified": verified["conv_bp_3x3_s2_c128_56"]}
                del sw, sgo, sgi
            except Exception as e:  # the conv leg must never take the bench line down
                extras["conv_fp_3x3_c64_56"] = "failed: %r" % (e,)
                verified["conv_fp_3x3_c64_56"] = False
        del xf

    if rank == 0 and not multi and not args.no_extras:
        sample = min(n, 1 << 28)
        hxs = x[:sample].cpu().numpy()
        orc.cast(hxs[: 1 << 24], args.mode)  # warm pass (threads, page tables)
        out = np.empty_like(hxs)
        out.fill(0)
        passes, cpu_s = 0, 0.0
        while passes < 4 or (cpu_s * orc.num_threads() < 15.0 and passes < 40):  # ~15+ core-seconds of work
            c0 = time.perf_counter()
            orc.cast(hxs, args.mode, out=out)
            cpu_s += time.perf_counter() - c0
            passes += 1
        step()
        torch.cuda.synchronize()
        chk = slice(0, 1 << 22)
        gpu_head = y[chk].cpu().numpy().view(np.uint32)
        same_head = bool(np.array_equal(gpu_head, out[chk].view(np.uint32)))
        port = {"value": round(sample * passes / cpu_s, 1), "unit": "elements/s",
                "cores": orc.num_threads(), "kind": "port",
                "sample": "%s over the first %d elements of the same tensor, %d timed passes (%.2f s wall) of "
                          "the OpenMP C oracle (oracle/fpemu_oracle.c), output pre-faulted" % (
                              args.mode, sample, passes, cpu_s),
                "gpu_matches_oracle_on_first_4Mi": same_head}
        cpu_baseline = reference_baseline(hxs[: 1 << 26], args.mode, gpu_head) or port
        if cpu_baseline is not port:
            cpu_baseline["port"] = port

    if rank == 0:
        if verified:
            extras["verified"] = verified
            extras["all_verified"] = all(verified.values())
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
            "roofline": roofline, "cpu_baseline": cpu_baseline,
            # the second per-GPU rate: the per-tensor-scaled step (amax + all-reduce(max) of one word + cast, 12 B/elem)
            "scaled_step": extras.get("scaled_per_tensor"), "extras": extras,
        }
        print(json.dumps(line), flush=True)
    if multi:
        dist.barrier()  # rank 0's extras run longer than the others': leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
