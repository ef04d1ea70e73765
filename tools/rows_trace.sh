#!/bin/bash
# rocprofv3 kernel durations of the rows kernel for a few 16-bit block sizes (tools/rows_one.py: 20 launches each).
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for spec in ${SPECS:-bf16:8 bf16:16 bf16:32 bf16:64 bf16:128 bf16:256 bf16:512 bf16:1024 bf16:4096 bf16:16384 bf16:65536 f32:8 f32:64 f32:1024 f32:8192 f32:16384 f32:32768 f32:65536}; do
  set -- ${spec/:/ }
  rm -rf /tmp/kt
  timeout -s KILL 90 rocprofv3 --kernel-trace --stats -d /tmp/kt -o p --output-format csv -- python3 $R/tools/rows_one.py $1 $2 ${MODE:-E4M3_RNE} > /tmp/kt.log 2>&1 || { tail -n 5 /tmp/kt.log; exit 1; }
  python3 - $1 $2 <<'PY'
import csv, glob, sys
f = glob.glob("/tmp/kt/**/*kernel_stats.csv", recursive=True)[0]
for row in csv.DictReader(open(f)):
    if "rows_" in row["Name"]:
        ns = float(row["AverageNs"]); bpe = 8 if sys.argv[1] == "f32" else 4
        print(sys.argv[1], "block", sys.argv[2], "calls", row["Calls"], "avg %.1f us  min %.1f  max %.1f  -> %.0f GB/s" % (ns / 1e3, float(row["MinNs"]) / 1e3, float(row["MaxNs"]) / 1e3, bpe * (1 << 26) / ns), flush=True)
PY
done
