#!/usr/bin/env bash
# Round-4 experiment batch 2 (GPU box): product time under load (profiling build), copy yardstick, new tests
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/r04/exp2; mkdir -p $O
P=$PWD/pantr-cdc2023-experiments_amd
PANTR_LIB_PATH=$P/lib/libpantr_mi355x_phase.so timeout -k 10 600 python scripts/product_time_under_load.py > $O/product_under_load.txt 2>&1
cat $O/product_under_load.txt | grep -v amdgpu.ids
timeout -k 10 120 python -c "
import sys; sys.path.insert(0,'pantr-cdc2023-experiments_amd')
import pantr_amd as pa
print('copy bandwidth GB/s:', [round(pa.copy_bandwidth(0)) for _ in range(3)])" > $O/copy_bw.txt 2>&1
cat $O/copy_bw.txt | grep -v amdgpu.ids
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_quadcopter.py tests/test_gpu_multirank.py -x -q -m gpu -k "deadline or residency or speculation_changes or rank" > $O/pytest_new.txt 2>&1
tail -5 $O/pytest_new.txt
echo finished
