#!/bin/bash
# Runs on the GPU box via gpurun: GPU parity tests, smoke, a short bench.  Each GPU step has
# its own time limit and the chain stops at the first failure.
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread \
    > gpurun_out/pytest_gpu.log 2>&1
rc=$?
tail -n 25 gpurun_out/pytest_gpu.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -u -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
rc=$?
tail -n 5 gpurun_out/smoke.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python -u bench.py --steps 20 --warmup 5 > gpurun_out/bench.log 2> gpurun_out/bench.err
rc=$?
tail -n 5 gpurun_out/bench.log; tail -n 5 gpurun_out/bench.err
exit $rc
