#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
for which in plain hooked; do
  OUT=gpurun_out/bertprof_$which
  rm -rf $OUT; mkdir -p $OUT
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 tools/bert_gpu_time.py $which \
    > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
  grep "per forward" $OUT/run.log
  python3 - $OUT <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
ours = [r for r in rows if any(k in r["Name"] for k in ("cast_kernel", "rows_", "amax_kernel", "zero_word"))]
print("  all kernels %.3f ms / forward (23 forwards); fpemu kernels %.3f ms / forward in %d launches / forward" % (
    tot / 23e6, sum(float(r["TotalDurationNs"]) for r in ours) / 23e6, sum(int(r["Calls"]) for r in ours) // 23))
for r in rows[:8]:
    print("  %-90s %6s %8.2f ms %8.1f us" % (r["Name"][:90], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
done
