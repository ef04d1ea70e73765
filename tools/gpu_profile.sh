#!/bin/bash
# rocprofv3 passes for bench.py on the GPU box: kernel-trace stats, then HBM counters in
# separate --pmc runs (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass).
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- \
    python3 bench.py --steps 20 --warmup 5 --no-extras > $OUT/stats_bench.log 2>&1 || { tail -5 $OUT/stats_bench.log; exit 1; }
timeout -s KILL 180 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- \
    python3 bench.py --steps 3 --warmup 1 --no-extras > $OUT/pmc_fetch.log 2>&1 || { tail -5 $OUT/pmc_fetch.log; exit 1; }
timeout -s KILL 180 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- \
    python3 bench.py --steps 3 --warmup 1 --no-extras > $OUT/pmc_write.log 2>&1 || { tail -5 $OUT/pmc_write.log; exit 1; }
# cmodel GEMM, 8192^3, both variants (the 'next' row's kernels)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/gemm_stats -- \
    python3 tools/gemm_bench.py 8192x8192x8192 nn > $OUT/gemm_stats.log 2>&1 || { tail -5 $OUT/gemm_stats.log; exit 1; }
find $OUT -name '*.csv' | head -20
