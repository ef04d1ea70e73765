#!/bin/bash
# The one driver of every GPU-box measurement in tools/ (run it through gpurun: `gpurun -- bash tools/gpu.sh TASK [ARGS]`).
# Each GPU step has its own time limit and a task stops at its first failure.  Output goes to gpurun_out/; what is to be
# judged is copied into profiles/ by hand (tools/summarize_prof.py for the bench profile).  tools/README.md lists the tasks.
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
S="python3 $R/tools/prof_summary.py"
FPEMU_KEYS="cast_kernel rows_ amax_kernel zero_word cast_multi amax_multi cast64"
SQ_A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
SQ_B="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES SQ_INSTS_MFMA"

stats() {   # stats <outdir> <seconds> <program...>: rocprofv3 --kernel-trace --stats
  local out=$1 t=$2; shift 2
  rm -rf $out; mkdir -p $out
  timeout -k 10 $t rocprofv3 --kernel-trace --stats --output-format csv -d $out -- "$@" > $out/run.log 2>&1 || { tail -n 5 $out/run.log; exit 1; }
}
pmc() {     # pmc <outdir> "<counters>" <program...>: one --pmc pass (at most 8 SQ / 4 TCC counters)
  local out=$1 ctr=$2; shift 2
  rm -rf $out; mkdir -p $out
  timeout -s KILL 180 rocprofv3 --pmc $ctr --output-format csv -d $out -- "$@" > $out/run.log 2>&1 || { tail -n 5 $out/run.log; exit 1; }
}

task=${1:-help}; shift || true
case $task in
check)      # the GPU suite, the 1-GPU bench line, the 2-rank gloo rehearsal of the bench (ranks share the GPU)
  timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/pytest_gpu.log 2>&1 \
    || { tail -n 40 gpurun_out/pytest_gpu.log; exit 1; }
  tail -n 3 gpurun_out/pytest_gpu.log
  timeout -k 10 400 python -u bench.py --steps 20 --warmup 5 > gpurun_out/bench.log 2> gpurun_out/bench.err \
    || { tail -n 30 gpurun_out/bench.err; exit 1; }
  python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/bench.log").read().strip().splitlines()[-1])
print("value %.4g elem/s  ms %.4f  frac %.4f  events frac %.4f" % (d["value"], d["ms_per_step"], d["hbm_frac_of_peak"], d["roofline"]["frac"]))
e = d["extras"]
print("ceilings", e.get("ceilings"))
print("placement", (e.get("placement") or {}).get("GB/s"))
print("verified", e.get("verified"), "all:", e.get("all_verified"))
for k, v in e.items():
    if isinstance(v, dict) and any(x in v for x in ("GB/s", "TFLOP/s", "GB/s_per_gpu", "TFLOP/s_effective")):
        print(k, v)
print("modes", e.get("modes_GB/s"))
print("cpu", {k: d["cpu_baseline"][k] for k in ("value", "kind", "cores")} if d.get("cpu_baseline") else None)
PY
  FPEMU_BENCH_BACKEND=gloo FPEMU_BENCH_WATCHDOG=240 timeout -k 10 300 python -u bench.py --gpus 2 --steps 10 --warmup 3 \
    > gpurun_out/bench_gloo2.log 2> gpurun_out/bench_gloo2.err || { tail -n 30 gpurun_out/bench_gloo2.err; exit 1; }
  python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_gloo2.log").read().strip().splitlines()[-1])
print("gloo x2 (one GPU): n_gpus", d["n_gpus"], "checksums", d["extras"].get("sharded_checksums"), "all_verified", d["extras"].get("all_verified"))
PY
  # the N > 1 flow's RCCL calls through a ONE-rank communicator (the only RCCL run a 1-GPU box allows)
  FPEMU_BENCH_REHEARSE=1 FPEMU_BENCH_WATCHDOG=240 timeout -k 10 300 python -u -m torch.distributed.run --nnodes=1 \
    --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 10 --warmup 3 \
    > gpurun_out/bench_rccl1.log 2> gpurun_out/bench_rccl1.err || { tail -n 30 gpurun_out/bench_rccl1.err; exit 1; }
  python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_rccl1.log").read().strip().splitlines()[-1])
print("rccl x1 rehearsal: scaled_step", d["scaled_step"], "checksums", d["extras"].get("sharded_checksums"), "all_verified", d["extras"].get("all_verified"))
PY
  ;;
profile)    # bench.py under rocprofv3: kernel stats, then FETCH_SIZE and WRITE_SIZE in separate passes; GEMM kernel stats
  P=gpurun_out/prof
  stats $P/stats 300 python3 bench.py --steps 20 --warmup 5 --no-extras
  pmc $P/pmc_fetch FETCH_SIZE python3 bench.py --steps 3 --warmup 1 --no-extras
  pmc $P/pmc_write WRITE_SIZE python3 bench.py --steps 3 --warmup 1 --no-extras
  stats $P/gemm_stats 300 python3 tools/gemm_bench.py 8192x8192x8192 nn
  python3 tools/summarize_prof.py $P ${1:-r03}
  ;;
tools)      # every tools/ benchmark in turn -> gpurun_out/tools.txt
  OUT=gpurun_out/tools.txt; : > $OUT
  run() { local t=$1; shift; echo "== $*" >> $OUT
          timeout -k 10 $t "$@" >> $OUT 2> gpurun_out/tools.err || { echo "FAILED: $*"; tail -n 20 gpurun_out/tools.err; exit 1; }
          echo "done: $*"; }
  run 120 python tools/scaled_bench.py
  run 120 python tools/storage_bench.py
  run 240 python tools/rows_bench.py
  run 120 python tools/multi_bench.py
  run 120 python tools/conv_bench.py
  run 240 python tools/gemm_bench.py
  run 200 python tools/out_amax_bench.py
  run 120 python tools/latency.py
  run 240 python tools/bert_overhead.py
  run 300 python tools/model_overhead.py
  if [ -x tools/mfma_zero ]; then run 60 tools/mfma_zero; fi
  ;;
model)      # per-kernel GPU time of the emulated ResNet-50 training step (13 steps)
  stats gpurun_out/rn 400 python3 tools/model_overhead.py emu-only
  tail -n 1 gpurun_out/rn/run.log
  $S kstats gpurun_out/rn $FPEMU_KEYS --per 13
  ;;
bert)       # BERT-base forward, plain and hooked: GPU kernel time against wall time (23 forwards)
  for which in plain hooked; do
    stats gpurun_out/bertprof_$which 300 python3 tools/bert_gpu_time.py $which
    grep "per forward" gpurun_out/bertprof_$which/run.log
    $S kstats gpurun_out/bertprof_$which $FPEMU_KEYS --per 23 --top 8
  done
  ;;
conv)       # one convolution shape of tools/conv_bench.py (default 1 = 3x3, 64 ch, 56x56): kernel stats + HBM counters
  SHAPE=${1:-1}; P=gpurun_out/convprof
  stats $P/stats 200 python3 tools/conv_bench.py $SHAPE
  pmc $P/pmc_fetch FETCH_SIZE python3 tools/conv_bench.py $SHAPE
  pmc $P/pmc_write WRITE_SIZE python3 tools/conv_bench.py $SHAPE
  $S kstats $P/stats; $S hbm $P
  ;;
conv-kernels) # per-kernel durations of several conv_bench shapes (SHAPES="0 1 5 6 9"): where the helpers stand beside the GEMMs
  for SHAPE in ${SHAPES:-0 1 5 6 9}; do
    P=gpurun_out/convk_$SHAPE
    stats $P 200 python3 tools/conv_bench.py $SHAPE
    echo "== shape $SHAPE: $(grep '^N' $P/run.log)"
    $S kstats $P --top 12
  done
  ;;
conv-pmc)   # SQ counters of the convolution's GEMM kernel (KERNEL=substring to pick another), three passes
  SHAPE=${1:-1}; P=gpurun_out/conv_pmc_$SHAPE
  pmc $P/a "$SQ_A" python3 tools/conv_bench.py $SHAPE
  pmc $P/b "$SQ_B" python3 tools/conv_bench.py $SHAPE
  pmc $P/c "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" python3 tools/conv_bench.py $SHAPE
  $S pmc $P ${KERNEL:-gemm_mfma_conv}
  ;;
gemm-pmc)   # SQ counters of a c-model GEMM kernel at 8192^3 (v0 = VALU separate, v1 = MFMA fmaf chain)
  V=${1:-v0}; P=gpurun_out/gemm_pmc_$V
  pmc $P/a "$SQ_A" python3 tools/gemm_bench.py 8192x8192x8192 nn $V
  pmc $P/b "${SQ_B/SQ_INSTS_MFMA/GRBM_GUI_ACTIVE}" python3 tools/gemm_bench.py 8192x8192x8192 nn $V
  grep -h "TFLOP" $P/a/run.log $P/b/run.log
  $S pmc $P gemm_
  ;;
rows-trace) # kernel durations of the rows kernels by storage and row length (SPECS="bf16:8 f32:65536 ...", MODE=)
  for spec in ${SPECS:-bf16:8 bf16:16 bf16:32 bf16:64 bf16:128 bf16:256 bf16:512 bf16:1024 bf16:4096 bf16:16384 bf16:65536 f32:8 f32:64 f32:1024 f32:8192 f32:16384 f32:32768 f32:65536}; do
    set -- ${spec/:/ }
    stats /tmp/kt 90 python3 $R/tools/rows_one.py $1 $2 ${MODE:-E4M3_RNE}
    echo "$1 rows of $2:"; $S kstats /tmp/kt rows_ --top 3 | grep "rows_"
  done
  ;;
rows-gaps)  # back-to-back rows launches: kernel duration and the gap to the next launch
  for spec in ${SPECS:-bf16:8 bf16:32 bf16:128 f32:32 f32:16384}; do
    set -- ${spec/:/ }
    rm -rf /tmp/kt
    timeout -s KILL 90 rocprofv3 --kernel-trace -d /tmp/kt -o p --output-format csv -- python3 $R/tools/rows_one.py $1 $2 ${MODE:-E4M3_RNE} > /tmp/kt.log 2>&1 || { tail -n 5 /tmp/kt.log; exit 1; }
    echo "$1 rows of $2:"; $S gaps /tmp/kt rows_
  done
  ;;
rows-pmc)   # SQ counters of the rows kernel for a few 16-bit block sizes
  for bs in ${SPECS:-8 32 128 1024}; do
    pmc /tmp/pmc_$bs "SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU" python3 $R/tools/rows_one.py bf16 $bs E4M3_RNE
    echo "bf16 block $bs:"; $S pmc /tmp/pmc_$bs rows_
  done
  ;;
scaled-gaps) # per-tensor quantisation back to back: zero / amax / cast durations and gaps (SPECS="1024:ws 512:ws")
  for spec in ${SPECS:-1024:ws 1024:static 512:ws}; do
    set -- ${spec/:/ }
    rm -rf /tmp/kt
    timeout -s KILL 120 rocprofv3 --kernel-trace -d /tmp/kt -o p --output-format csv -- python3 $R/tools/scaled_one.py $1 $2 > /tmp/kt.log 2>&1 || { tail -n 5 /tmp/kt.log; exit 1; }
    grep "per call" /tmp/kt.log
    $S gaps /tmp/kt zero_word amax_kernel cast_kernel
  done
  ;;
build-sweep) # CPU side: tools/sweep (+ sweep_pl with kernarg preload) against the in-tree library
  cd $R; L=fp8-emulation-toolkit_amd/mpemu/pytquant/hip
  F="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -L$L -lfpemu_hip -Wl,-rpath,\$ORIGIN/../$L"
  /opt/rocm/bin/hipcc $F -o tools/sweep tools/sweep.hip
  /opt/rocm/bin/hipcc $F -mllvm -amdgpu-kernarg-preload-count=8 -o tools/sweep_pl tools/sweep.hip
  ;;
*)
  grep -E "^[a-z-]+\\) +#" "$0" | sed -E "s/\\) +#/:/"
  ;;
esac
