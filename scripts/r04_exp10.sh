#!/usr/bin/env bash
# Phase timing and the sixteen-piece timing of the time outside the evaluations, all three problems (profiling builds)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/r04/exp10; mkdir -p $O
P=$PWD/pantr-cdc2023-experiments_amd
PANTR_LIB_PATH=$P/lib/libpantr_mi355x_phase.so timeout -k 10 300 python scripts/phase_timing.py 2>&1 | grep -v amdgpu.ids > $O/phase.txt || exit 1
cat $O/phase.txt
bash scripts/r04_exp8.sh "$@"
