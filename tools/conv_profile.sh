#!/bin/bash
# rocprofv3 kernel summary + HBM counters (separate --pmc passes) for one convolution shape of tools/conv_bench.py
# (default 1 = ResNet-50 3x3, 64 channels, 56 x 56, batch 32).  Output: gpurun_out/convprof/
set -o pipefail
export TMPDIR=/tmp
SHAPE=${1:-1}
OUT=gpurun_out/convprof
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 tools/conv_bench.py $SHAPE \
  > $OUT/stats.log 2>&1 || { tail -5 $OUT/stats.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 tools/conv_bench.py $SHAPE \
  > $OUT/pmc_fetch.log 2>&1 || { tail -5 $OUT/pmc_fetch.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 tools/conv_bench.py $SHAPE \
  > $OUT/pmc_write.log 2>&1 || { tail -5 $OUT/pmc_write.log; exit 1; }
python3 - <<PY
import csv, glob, collections
st = sorted(glob.glob("$OUT/stats/*/*_kernel_stats.csv"))[-1]
print("| kernel | calls | avg us | % |")
for r in csv.DictReader(open(st)):
    print("| %s | %s | %.1f | %s |" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
for which, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    f = sorted(glob.glob("$OUT/%s/*/*_counter_collection.csv" % which))[-1]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == ctr:
            acc[r["Kernel_Name"][:90]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print("%s %s: avg %.1f KiB over %d launches" % (ctr, k, sum(v) / len(v), len(v)))
PY
