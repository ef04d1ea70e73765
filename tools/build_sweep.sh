#!/bin/bash
# Builds tools/sweep (and tools/sweep_pl: same kernels with gfx950 kernarg preload) against the in-tree library.
set -e
cd "$(dirname "$0")/.."
L=fp8-emulation-toolkit_amd/mpemu/pytquant/hip
F="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -L$L -lfpemu_hip -Wl,-rpath,\$ORIGIN/../$L"
/opt/rocm/bin/hipcc $F -o tools/sweep tools/sweep.hip
/opt/rocm/bin/hipcc $F -mllvm -amdgpu-kernarg-preload-count=8 -o tools/sweep_pl tools/sweep.hip
