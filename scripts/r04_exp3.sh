#!/usr/bin/env bash
# Round-4 experiment batch 3: LDS-DMA offset semantics; prefetch depth 2 (variant library) against depth 1
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/r04/exp3; mkdir -p $O
P=$PWD/pantr-cdc2023-experiments_amd
( cd scripts/microbench && timeout -k 10 60 ./lds_dma_offset_test ) > $O/lds_dma_offset.txt 2>&1
cat $O/lds_dma_offset.txt
for lib in libpantr_mi355x.so libpantr_d2.so; do
  export PANTR_LIB_PATH=$P/lib/$lib
  echo "== $lib" >> $O/ab.txt
  timeout -k 10 200 python scripts/gpu_timing.py 2>&1 | grep -v amdgpu.ids | grep "both600\|cold300" >> $O/ab.txt
  timeout -k 10 300 python scripts/traffic_workload.py tile8 2>&1 | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('tile8', [round(x,1) for x in d['kernel_ms']], round(d['steps_per_s']))" >> $O/ab.txt
  timeout -k 10 300 python scripts/traffic_workload.py stress 2>&1 | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('stress', [round(x,1) for x in d['kernel_ms']], round(d['steps_per_s']))" >> $O/ab.txt
done
cat $O/ab.txt
PANTR_LIB_PATH=$P/lib/libpantr_d2.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/pytest_d2.txt 2>&1
tail -3 $O/pytest_d2.txt
echo finished
