#!/usr/bin/env bash
# A/B on one box: fused passes (tree) vs the previous pantr_inner (libpantr_nofuse.so, PANTR_ALLOW_STALE_LIB)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/r04/exp7; mkdir -p $O; rm -f $O/ab.txt
P=$PWD/pantr-cdc2023-experiments_amd
export PANTR_ALLOW_STALE_LIB=1
for round in 1 2 3; do
for lib in libpantr_mi355x.so libpantr_nofuse.so; do
  export PANTR_LIB_PATH=$P/lib/$lib
  echo "== $lib (round $round)" >> $O/ab.txt
  timeout -k 10 200 python scripts/gpu_timing.py 2>&1 | grep -v amdgpu.ids | grep "both600\|cold300" | cut -c1-70 >> $O/ab.txt
done
done
for lib in libpantr_mi355x.so libpantr_nofuse.so; do
  export PANTR_LIB_PATH=$P/lib/$lib
  echo "== $lib quad" >> $O/ab.txt
  timeout -k 10 300 python scripts/quad_cold_timing.py 2>&1 | grep -v amdgpu.ids >> $O/ab.txt
done
cat $O/ab.txt
