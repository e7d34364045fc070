#!/usr/bin/env bash
# Per-phase solve times of profiling builds of library variants (built here or on the CPU box with
#   make phase PHASE_LIB=<file> EXTRA_HIPFLAGS=-D...); results to gpurun_out/ab/<name>.txt.
#   scripts/ab_phase.sh lib/libpantr_var_x_phase.so ... [-- problems for phase_timing.py]
set -e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
P=$PWD/pantr-cdc2023-experiments_amd
mkdir -p gpurun_out/ab
libs=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do libs+=("$1"); shift; done
[ "$1" = "--" ] && shift
for lib in "${libs[@]}"; do
    name=$(basename "$lib" .so)
    PANTR_LIB_PATH=$P/$lib timeout -k 10 300 python scripts/phase_timing.py "${@:-hanging_chain}" > gpurun_out/ab/$name.txt 2>&1
    grep -v amdgpu.ids gpurun_out/ab/$name.txt
done
