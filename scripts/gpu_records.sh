#!/usr/bin/env bash
# The records of one round on the GPU box (run through gpurun), all written under gpurun_out/<tag>/:
#   rocprofv3 kernel trace + PMC passes of the headline bench (scripts/profile.sh -> gpurun_out/prof),
#   the kernel trace of the other solve kernels (scripts/rows_workload.py),
#   the batch tiled R times (scripts/saturation_timing.py),
#   the sharded bench mode rehearsed with two ranks on the one GPU (gloo, PANTR_BENCH_REHEARSAL=1).
#   scripts/gpu_records.sh <tag>
set -e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
TAG=${1:-r03}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout -k 10 400 bash scripts/profile.sh > $OUT/profile.log 2>&1
rm -rf gpurun_out/prof_rows
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_rows -- python3 scripts/rows_workload.py > $OUT/rows_workload.txt 2>&1
timeout -k 10 120 python scripts/saturation_timing.py 1 2 4 8 > $OUT/saturation.txt 2>&1
for W in sweep stress; do
  PANTR_BENCH_REHEARSAL=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
      --master-port 29571 bench.py --gpus 2 --steps 3 --warmup 1 --workload $W --scaling strong > $OUT/rehearsal_strong_$W.log 2>&1
done
PANTR_BENCH_REHEARSAL=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
    --master-port 29572 bench.py --gpus 2 --steps 5 --warmup 2 > $OUT/rehearsal_weak.log 2>&1
timeout -k 10 200 python bench.py --steps 10 --warmup 2 --streams 4 --headline-only > $OUT/bench_streams4.json 2> $OUT/bench_streams4.err
echo done > $OUT/done
