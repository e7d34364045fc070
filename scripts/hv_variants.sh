#!/usr/bin/env bash
# Timing-only variants of the library (wrong results by construction): how much of a hanging-chain
# Hessian-vector product is cross-lane traffic, how much is waiting for the tape DMA?
#   scripts/hv_variants.sh        (on a GPU box; needs hipcc)
set -e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
P=pantr-cdc2023-experiments_amd
cp $P/lib/libpantr_mi355x.so /tmp/pantr_ok.so
trap 'cp /tmp/pantr_ok.so '"$P"'/lib/libpantr_mi355x.so' EXIT
echo "== baseline"; timeout -k 5 100 python scripts/hv_timing.py 600
for v in NO_SHUFFLE NO_VMWAIT; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -fPIC -shared -DPANTR_TIMING_$v \
      -o $P/lib/libpantr_mi355x.so $P/csrc/pantr_capi.hip
  echo "== $v"; timeout -k 5 100 python scripts/hv_timing.py 600
done
