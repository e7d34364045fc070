#!/usr/bin/env bash
# A/B on one box: the hanging chain's PANTR kernels with tree vs a variant library lib/libpantr_nb.so (make LIB=... EXTRA_HIPFLAGS=-D...)
# (libpantr_nb.so: -DPANTR_CHAIN_GENERIC_PASSES; kernel names differ, so PANTR_ALLOW_STALE_LIB is not needed)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/r04/exp9; mkdir -p $O; rm -f $O/ab.txt
P=$PWD/pantr-cdc2023-experiments_amd
for round in 1 2; do
for lib in libpantr_mi355x.so libpantr_nb.so; do
  export PANTR_LIB_PATH=$P/lib/$lib
  echo "== $lib (round $round)" >> $O/ab.txt
  timeout -k 10 200 python scripts/gpu_timing.py 2>&1 | grep -v amdgpu.ids | grep "both600\|cold300\|warm300" | cut -c1-200 >> $O/ab.txt
done
done
cat $O/ab.txt
