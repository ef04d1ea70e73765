#!/bin/bash
# One GPU-box call: the GPU suite, the 1-GPU bench line, the 2-rank gloo rehearsal of the bench (ranks share the
# GPU; bench.py launches its own ranks), (tools/gloo_probe.py, the demonstration of torch's gloo device-tensor hang, is run by hand: it ends in its watchdog by design).  Every GPU step has its own time limit and the chain
# stops at the first failure.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/pytest_gpu.log 2>&1 \
  || { tail -n 40 gpurun_out/pytest_gpu.log; exit 1; }
tail -n 3 gpurun_out/pytest_gpu.log
timeout -k 10 400 python -u bench.py --steps 20 --warmup 5 > gpurun_out/bench.log 2> gpurun_out/bench.err \
  || { tail -n 30 gpurun_out/bench.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench.log").read().strip().splitlines()[-1])
print("value %.4g elem/s  ms %.4f  frac %.4f  events frac %.4f" % (d["value"], d["ms_per_step"], d["hbm_frac_of_peak"], d["roofline"]["frac"]))
e = d["extras"]
print("ceilings", e.get("ceilings"))
print("verified", e.get("verified"), "all:", e.get("all_verified"))
for k, v in e.items():
    if isinstance(v, dict) and ("GB/s" in v or "TFLOP/s" in v or "GB/s_per_gpu" in v or "TFLOP/s_effective" in v):
        print(k, v)
print("modes", e.get("modes_GB/s"))
print("cpu", {k: d["cpu_baseline"][k] for k in ("value", "kind", "cores")} if d.get("cpu_baseline") else None)
PY
FPEMU_BENCH_BACKEND=gloo FPEMU_BENCH_WATCHDOG=240 timeout -k 10 300 python -u bench.py --gpus 2 --steps 10 --warmup 3 \
  > gpurun_out/bench_gloo2.log 2> gpurun_out/bench_gloo2.err || { tail -n 30 gpurun_out/bench_gloo2.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_gloo2.log").read().strip().splitlines()[-1])
print("gloo x2 (one GPU): n_gpus", d["n_gpus"], "checksums", d["extras"].get("sharded_checksums"), "all_verified", d["extras"].get("all_verified"))
PY
