#!/bin/bash
# Per-tensor quantisation back to back under rocprofv3 --kernel-trace: every kernel's duration and the gap between a
# kernel's end and the next one's start, averaged per position in the call's launch sequence.
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for spec in ${SPECS:-1024:ws 1024:static 512:ws}; do
  set -- ${spec/:/ }
  rm -rf /tmp/kt
  timeout -s KILL 120 rocprofv3 --kernel-trace -d /tmp/kt -o p --output-format csv -- python3 $R/tools/scaled_one.py $1 $2 > /tmp/kt.log 2>&1 || { tail -n 5 /tmp/kt.log; exit 1; }
  grep "per call" /tmp/kt.log
  python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/kt/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if any(k in r["Kernel_Name"] for k in ("amax_kernel", "cast_kernel", "zero_word"))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) // 2:]           # the timed half
names = [next(k for k in ("amax_kernel", "cast_kernel", "zero_word") if k in r["Kernel_Name"]) for r in rows]
first = names.index("cast_kernel") + 1      # start at a call boundary
rows, names = rows[first:], names[first:]
per = names.index("cast_kernel") + 1
acc = {}
for i, r in enumerate(rows):
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    g = (int(rows[i + 1]["Start_Timestamp"]) - int(r["End_Timestamp"])) / 1e3 if i + 1 < len(rows) else None
    a = acc.setdefault((i % per, names[i]), [[], []])
    a[0].append(d)
    if g is not None:
        a[1].append(g)
tot = 0.0
for (pos, nm), (d, g) in sorted(acc.items()):
    print("  %d %-14s dur %8.1f us   gap to next %6.1f us" % (pos, nm, sum(d) / len(d), sum(g) / max(1, len(g))))
    tot += sum(d) / len(d) + sum(g) / max(1, len(g))
print("  per call: %.1f us" % tot, flush=True)
PY
done
