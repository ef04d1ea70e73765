#!/bin/bash
# One GPU-box call: every tools/ benchmark in turn, output gathered in gpurun_out/tools.txt (copied to
# profiles/tools_rNN.txt by hand).  Each step has its own time limit; the chain stops at the first failure.
set -o pipefail
mkdir -p gpurun_out
OUT=gpurun_out/tools.txt
: > $OUT
run() {  # run <seconds> <command...>
  local t=$1; shift
  echo "== $*" >> $OUT
  timeout -k 10 $t "$@" >> $OUT 2> gpurun_out/tools.err || { echo "FAILED: $*"; tail -n 20 gpurun_out/tools.err; exit 1; }
  echo "done: $*"
}
run 120 python tools/scaled_bench.py
run 120 python tools/storage_bench.py
run 240 python tools/rows_bench.py
run 120 python tools/multi_bench.py
run 120 python tools/conv_bench.py
run 240 python tools/gemm_bench.py
run 120 python tools/latency.py
run 240 python tools/bert_overhead.py
run 300 python tools/model_overhead.py
run 60 tools/mfma_zero
