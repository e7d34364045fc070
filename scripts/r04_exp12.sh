#!/usr/bin/env bash
# A/B on one box: shards of the N=120 stress batch (128 / 256 / 512 instances) with the compile-time passes (tree)
# and with the generic ones (lib/libpantr_nb.so built with -DPANTR_CHAIN_GENERIC_PASSES)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
P=$PWD/pantr-cdc2023-experiments_amd
for round in 1 2; do
for lib in libpantr_mi355x.so libpantr_nb.so; do
  PANTR_LIB_PATH=$P/lib/$lib timeout -k 10 300 python scripts/stress_shard_timing.py 2>&1 | grep -v amdgpu.ids
done
done
