#!/bin/bash
# Back-to-back launches of one rows case under rocprofv3 --kernel-trace: kernel durations AND the gaps between a
# kernel's end and the next one's start (what an event pair around N launches adds to the kernel time).
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for spec in ${SPECS:-bf16:8 bf16:32 bf16:128 f32:32 f32:16384}; do
  set -- ${spec/:/ }
  rm -rf /tmp/kt
  timeout -s KILL 90 rocprofv3 --kernel-trace -d /tmp/kt -o p --output-format csv -- python3 $R/tools/rows_one.py $1 $2 ${MODE:-E4M3_RNE} > /tmp/kt.log 2>&1 || { tail -n 5 /tmp/kt.log; exit 1; }
  python3 - $1 $2 <<'PY'
import csv, glob, sys
f = glob.glob("/tmp/kt/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "rows_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
gap = [(int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3 for a, b in zip(rows, rows[1:])]
print(sys.argv[1], "block", sys.argv[2], "n", len(rows), "dur avg %.1f us" % (sum(dur) / len(dur)), "gaps:", " ".join("%.1f" % g for g in gap), flush=True)
PY
done
