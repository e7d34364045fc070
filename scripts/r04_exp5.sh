#!/usr/bin/env bash
# Round-4 batch 5: A/B of library variants on one box (headline batch, tile8, stress), three rounds each
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/r04/exp5; mkdir -p $O; rm -f $O/ab.txt
P=$PWD/pantr-cdc2023-experiments_amd
for round in 1 2; do
for lib in "$@"; do
  export PANTR_LIB_PATH=$P/lib/$lib
  echo "== $lib (round $round)" >> $O/ab.txt
  timeout -k 10 200 python scripts/gpu_timing.py 2>&1 | grep -v amdgpu.ids | grep "both600" | cut -c1-60 >> $O/ab.txt
  for w in tile8 stress; do
  timeout -k 10 300 python scripts/traffic_workload.py $w 2>&1 | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['workload'], [round(x,1) for x in d['kernel_ms']], round(d['steps_per_s']))" >> $O/ab.txt
  done
done
done
cat $O/ab.txt
echo finished
