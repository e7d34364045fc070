#!/usr/bin/env bash
# Where does solve time go?  Builds the library with -DPANTR_PHASE_TIMING (stats fields final_* are
# repurposed as phase timers; see pantr_kernel.hip.h) into a scratch copy and prints per-phase means.
#   scripts/phase_timing.sh            (on a GPU box; needs hipcc)
set -e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
P=pantr-cdc2023-experiments_amd
cp $P/lib/libpantr_mi355x.so /tmp/pantr_ok.so
trap 'cp /tmp/pantr_ok.so '"$P"'/lib/libpantr_mi355x.so' EXIT
/opt/rocm/bin/hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -fPIC -shared -DPANTR_PHASE_TIMING \
    -o $P/lib/libpantr_mi355x.so $P/csrc/pantr_capi.hip
timeout -k 10 300 python scripts/phase_timing.py "$@"
