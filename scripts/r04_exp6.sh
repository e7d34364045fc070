#!/usr/bin/env bash
# Round-4 batch 6: fused element-wise passes of pantr_inner - GPU suite, timings of every configuration
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/r04/exp6; mkdir -p $O
P=$PWD/pantr-cdc2023-experiments_amd
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1
tail -4 $O/pytest_gpu.txt
timeout -k 10 200 python scripts/gpu_timing.py 2>&1 | grep -v amdgpu.ids | grep "both600" | cut -c1-70 > $O/ab.txt
for w in tile8 stress; do
timeout -k 10 300 python scripts/traffic_workload.py $w 2>&1 | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['workload'], [round(x,1) for x in d['kernel_ms']], round(d['steps_per_s']))" >> $O/ab.txt
done
cat $O/ab.txt
PANTR_LIB_PATH=$P/lib/libpantr_mi355x_phase.so timeout -k 10 300 python scripts/phase_timing.py 2>&1 | grep -v amdgpu.ids > $O/phase_timing.txt
cat $O/phase_timing.txt
timeout -k 10 300 python scripts/quad_straggler.py quadcopter 2>&1 | grep -v amdgpu.ids | tail -4
timeout -k 10 300 python scripts/quad_straggler.py realistic_quadcopter 2>&1 | grep -v amdgpu.ids | tail -4
echo finished
