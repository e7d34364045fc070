#!/usr/bin/env bash
# Round-4 batch 4: the tree's library (one-wavefront kernel two stages ahead, two-wavefront kernel one) - timings, full GPU suite
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/r04/exp4; mkdir -p $O
P=$PWD/pantr-cdc2023-experiments_amd
timeout -k 10 200 python scripts/gpu_timing.py 2>&1 | grep -v amdgpu.ids | grep "both600\|cold300\|warm300" > $O/ab.txt
for w in tile8 tile4 stress; do
timeout -k 10 300 python scripts/traffic_workload.py $w 2>&1 | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['workload'], [round(x,1) for x in d['kernel_ms']], round(d['steps_per_s']), d['kernel'])" >> $O/ab.txt
done
cat $O/ab.txt
PANTR_LIB_PATH=$P/lib/libpantr_mi355x_phase.so timeout -k 10 600 python scripts/product_time_under_load.py 2>&1 | grep -v amdgpu.ids > $O/product_under_load.txt
cat $O/product_under_load.txt
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1
tail -5 $O/pytest_gpu.txt
echo finished
