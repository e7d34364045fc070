#!/usr/bin/env bash
# A/B on one box: a variant library built from the tree (lib/libpantr_nb.so) against the library of the last commit
# (lib/libpantr_mi355x.so, PANTR_ALLOW_STALE_LIB=1): the hanging chain's parity tests on the variant, then pass times
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/r04/exp11; mkdir -p $O; rm -f $O/ab.txt
P=$PWD/pantr-cdc2023-experiments_amd
export PANTR_ALLOW_STALE_LIB=1
PANTR_LIB_PATH=$P/lib/libpantr_nb.so timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -n 3 | tee -a $O/ab.txt
for round in 1 2; do
for lib in libpantr_mi355x.so libpantr_nb.so; do
  export PANTR_LIB_PATH=$P/lib/$lib
  echo "== $lib (round $round)" >> $O/ab.txt
  timeout -k 10 200 python scripts/gpu_timing.py 2>&1 | grep -v amdgpu.ids | grep "both600\|cold300" | cut -c1-160 >> $O/ab.txt
  timeout -k 10 200 python scripts/saturation_timing.py 8 2>&1 | grep -v amdgpu.ids | cut -c1-120 >> $O/ab.txt
done
done
cat $O/ab.txt
