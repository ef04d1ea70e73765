#!/bin/bash
# Per-kernel GPU time of the emulated ResNet-50 training step (tools/model_overhead.py emu-only).
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/rn
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- \
    python3 tools/model_overhead.py emu-only > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
tail -1 $OUT/run.log
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/rn/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("all kernels: %.1f ms over 13 steps" % (tot / 1e6))
ours = [r for r in rows if any(k in r["Name"] for k in ("cast_kernel", "rows_", "amax_kernel", "zero_word"))]
print("fpemu kernels: %.2f ms, %d calls" % (sum(float(r["TotalDurationNs"]) for r in ours) / 1e6,
                                            sum(int(r["Calls"]) for r in ours)))
for r in rows[:25]:
    print("%-100s %6s %8.2f ms %8.1f us" % (r["Name"][:100], r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                           float(r["AverageNs"]) / 1e3))
PY
