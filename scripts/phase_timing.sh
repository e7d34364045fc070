#!/usr/bin/env bash
# Where does solve time go?  Builds the profiling variant of the library (`make phase`:
# -DPANTR_PHASE_TIMING, written to lib/libpantr_mi355x_phase.so, never over the product library) and
# runs scripts/phase_timing.py against it through PANTR_LIB_PATH.
#   scripts/phase_timing.sh [hanging_chain|quadcopter|realistic_quadcopter|panoc|strucpanoc ...]
set -e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
P=pantr-cdc2023-experiments_amd
make -s -C $P phase
PANTR_LIB_PATH=$PWD/$P/lib/libpantr_mi355x_phase.so timeout -k 10 300 python scripts/phase_timing.py "$@"
