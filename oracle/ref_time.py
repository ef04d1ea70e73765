"""Time the reference's own CPU extension (oracle/_ref/native, built by oracle/build_ref.py from the
sources under /root/reference in the build container) on this host's cores.

Test infrastructure only: run as a CHILD process by bench.py's `cpu_baseline` leg, so that a host whose
CPU lacks an instruction the -march=native build uses ends this process (SIGILL) and not the bench --
bench.py then falls back to the C restatement ("port").  Nothing here reads /root/reference: only the
prebuilt extension travels to the GPU box.

    python oracle/ref_time.py IN.npy OUT_HEAD.npy MODE SECONDS

IN.npy: fp32 sample.  Calls the reference's pybind entry point exactly as its operator does,
fpemu_cpp.forward(input, mode, size, inplace, scale, block_norm, block_size)  (pytquant/cpp/fpemu.py:59-61,
cpp/fpemu_impl.cpp:2013-2028), out-of-place (the reference allocates zeros_like) and in-place, each for about
SECONDS/2 of wall time, writes the first 4 Mi outputs to OUT_HEAD.npy and prints one JSON line.
"""
import importlib.util
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "_ref", "native", "fpemu_ref_native.so")


def load_ext():
    import torch  # noqa: F401  (the extension links against libtorch)

    spec = importlib.util.spec_from_file_location("fpemu_ref_native", SO)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    src, dst, mode, budget = sys.argv[1], sys.argv[2], sys.argv[3], float(sys.argv[4])
    import numpy as np
    import torch

    ext = load_ext()
    x = torch.from_numpy(np.load(src))
    n = x.numel()
    ext.forward(x[: 1 << 20].clone(), mode, 1 << 20, False, 1.0, False, 1)  # warm: threads, code pages

    def run(inplace):
        passes, spent = 0, 0.0
        out = None
        while passes < 2 or (spent < budget / 2 and passes < 2000):
            t = x.clone() if inplace else x
            t0 = time.perf_counter()
            out = ext.forward(t, mode, n, inplace, 1.0, False, 1)[0]
            spent += time.perf_counter() - t0
            passes += 1
        return passes, spent, out

    p_out, s_out, out = run(False)
    p_in, s_in, _ = run(True)
    np.save(dst, out[: 1 << 22].numpy())
    print(json.dumps({"elements": n, "mode": mode,
                      "out_of_place": {"passes": p_out, "seconds": s_out, "elements_per_s": n * p_out / s_out},
                      "in_place": {"passes": p_in, "seconds": s_in, "elements_per_s": n * p_in / s_in},
                      "torch_threads": torch.get_num_threads()}), flush=True)


if __name__ == "__main__":
    main()
