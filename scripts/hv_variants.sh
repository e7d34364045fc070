#!/usr/bin/env bash
# timing-only variants of the library (wrong results by construction): where does Hv time go?
cd "${GRAFT_REPO_ROOT:-.}"
L=pantr-cdc2023-experiments_amd/lib
cp $L/libpantr_mi355x.so /tmp/ok.so
echo "== baseline"; timeout -k 5 100 python scripts/hv_timing.py 600
for v in NO_SHUFFLE NO_VMWAIT; do
  cp $L/libpantr_$v.so $L/libpantr_mi355x.so
  echo "== $v"; timeout -k 5 100 python scripts/hv_timing.py 600
done
cp /tmp/ok.so $L/libpantr_mi355x.so
