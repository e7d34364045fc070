#!/usr/bin/env bash
# The time of a PANTR iteration outside the evaluations in sixteen pieces (four builds, -DPANTR_REST_SPLIT=0|4|8|12)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/r04/exp8; mkdir -p $O; rm -f $O/split.txt
P=$PWD/pantr-cdc2023-experiments_amd
for b in 0 4 8 12; do
  export PANTR_LIB_PATH=$P/lib/libpantr_split$b.so
  timeout -k 10 250 python scripts/rest_timing.py "$@" 2>&1 | grep -v amdgpu.ids >> $O/split.txt || exit 1
done
cat $O/split.txt
