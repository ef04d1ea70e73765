#!/bin/bash
# SQ counters of the implicit convolution's GEMM kernel for one shape of tools/conv_bench.py (default 1 = 3x3, 64 ch, 56x56).
# Two --pmc passes of at most 8 SQ counters each.
set -o pipefail
export TMPDIR=/tmp
SHAPE=${1:-1}
OUT=gpurun_out/conv_pmc_$SHAPE
rm -rf $OUT; mkdir -p $OUT
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS \
  --output-format csv -d $OUT/a -- python3 tools/conv_bench.py $SHAPE > $OUT/a.log 2>&1 || { tail -5 $OUT/a.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES SQ_INSTS_MFMA \
  --output-format csv -d $OUT/b -- python3 tools/conv_bench.py $SHAPE > $OUT/b.log 2>&1 || { tail -5 $OUT/b.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE \
  --output-format csv -d $OUT/c -- python3 tools/conv_bench.py $SHAPE > $OUT/c.log 2>&1 || { tail -5 $OUT/c.log; }
python3 - $OUT <<'PY'
import csv, glob, sys, collections, os
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if os.environ.get("KERNEL", "gemm_mfma_conv") in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
v = {k: sum(x) / len(x) for k, x in acc.items()}
for k in sorted(v):
    print("%-26s %.4g per launch (%d launches)" % (k, v[k], len(acc[k])))
w = v.get("SQ_WAVE_CYCLES")
if w:
    for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS"):
        if k in v:
            print("%-26s %.1f %% of wave cycles" % (k, 100 * v[k] / w))
if v.get("SQ_LDS_IDX_ACTIVE"):
    print("LDS bank conflicts         %.2f %% of LDS-array cycles" % (100 * v.get("SQ_LDS_BANK_CONFLICT", 0) / v["SQ_LDS_IDX_ACTIVE"]))
PY
