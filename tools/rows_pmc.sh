#!/bin/bash
# SQ counters of the rows kernel for a few block sizes (one pass per case; <= 8 SQ counters per pass).
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for bs in 8 32 128 1024; do
  rm -rf /tmp/pmc_$bs
  timeout -s KILL 90 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU \
    -d /tmp/pmc_$bs -o p --output-format csv -- python3 $R/tools/rows_one.py bf16 $bs E4M3_RNE > /tmp/pmc_$bs.log 2>&1 || { tail -n 5 /tmp/pmc_$bs.log; exit 1; }
  python3 - $bs <<'PY'
import csv, glob, sys, collections
bs = sys.argv[1]
f = glob.glob("/tmp/pmc_%s/**/*counter_collection.csv" % bs, recursive=True)[0]
acc = collections.defaultdict(lambda: [0.0, 0])
for row in csv.DictReader(open(f)):
    if "rows_" not in row["Kernel_Name"]:
        continue
    a = acc[row["Counter_Name"]]
    a[0] += float(row["Counter_Value"]); a[1] += 1
print("bf16 block", bs, {k: round(v[0] / v[1]) for k, v in sorted(acc.items())}, flush=True)
PY
done
