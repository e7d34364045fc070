#!/usr/bin/env bash
# Round-4 experiment batch 1 (GPU box): layout A/B microbenchmark, residency caps in the memory-bound
# regimes, PMC traffic of the stress and tile8 workloads.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/r04/exp1; mkdir -p $O
set -x
( cd scripts/microbench && timeout -k 10 120 ./hv_layout_ab 50 ) > $O/hv_layout_ab.txt 2>&1
timeout -k 10 200 python scripts/hv_timing.py 1 > $O/hv_timing_1.txt 2>&1
timeout -k 10 200 python scripts/gpu_timing.py > $O/gpu_timing.txt 2>&1
for r in auto 2 3 4 5 6; do
  timeout -k 10 300 python scripts/traffic_workload.py tile8 residency=$r >> $O/residency_tile8.txt 2>&1 || break
done
for r in auto 1 2 3 4; do
  timeout -k 10 300 python scripts/traffic_workload.py stress residency=$r >> $O/residency_stress.txt 2>&1 || break
done
timeout -k 10 900 scripts/traffic_profile.sh stress tile8 > $O/traffic_profile.txt 2>&1

timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "deadline or residency or speculation_changes or solve_iterates or bench_workload" > $O/pytest_new.txt 2>&1
tail -3 $O/pytest_new.txt
echo finished
