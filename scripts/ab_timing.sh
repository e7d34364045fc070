#!/usr/bin/env bash
# A/B timing of library variants built from the SAME tree (make LIB=<file> EXTRA_HIPFLAGS=-D...):
# for each library given, the isolated Hessian-vector product (one wavefront on an empty device and
# 600 at once), the pass time of the bench batch and - with PARITY=1 - the GPU parity tests of the
# hanging chain.  Development aid; results go to gpurun_out/ab/<name>.txt.
#   scripts/ab_timing.sh lib/libpantr_mi355x.so lib/libpantr_var_x.so ...
set -e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
P=$PWD/pantr-cdc2023-experiments_amd
mkdir -p gpurun_out/ab
for lib in "$@"; do
    name=$(basename "$lib" .so)
    out=gpurun_out/ab/$name.txt
    export PANTR_LIB_PATH=$P/$lib
    {
        echo "== $name"
        timeout -k 10 120 python scripts/hv_timing.py 1
        timeout -k 10 120 python scripts/hv_timing.py 600
        timeout -k 10 200 python scripts/gpu_timing.py
    } > "$out" 2>&1
    if [ -n "$PARITY" ]; then
        timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu >> "$out" 2>&1
    fi
    grep -v amdgpu.ids "$out"
done
