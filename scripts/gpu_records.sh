#!/usr/bin/env bash
# The records of one round on the GPU box (run through gpurun), all written under gpurun_out/<tag>/:
#   the GPU test suite, the bench line,
#   rocprofv3 kernel trace + PMC passes of the headline bench (scripts/profile.sh -> gpurun_out/prof),
#   kernel trace + FETCH_SIZE / WRITE_SIZE passes of the memory-bound workloads (scripts/traffic_profile.sh),
#   the kernel trace of the other solve kernels (scripts/rows_workload.py),
#   the batch tiled R times (scripts/saturation_timing.py), per-phase times (profiling build),
#   the multi-rank bench modes started bare (bench.py launches its own ranks) with two ranks on the one GPU
#   (gloo, PANTR_BENCH_REHEARSAL=1).
#   scripts/gpu_records.sh <tag>
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
TAG=${1:-r04}
OUT=gpurun_out/$TAG
P=$PWD/pantr-cdc2023-experiments_amd
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -q -m gpu > $OUT/gpu_tests.log 2>&1; tail -2 $OUT/gpu_tests.log
( time timeout -k 10 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2> $OUT/bench_time.txt; tail -c 300 $OUT/bench.json; echo; cat $OUT/bench_time.txt
timeout -k 10 400 bash scripts/profile.sh > $OUT/profile.log 2>&1
timeout -k 10 600 bash scripts/traffic_profile.sh stress tile8 > $OUT/traffic_profile.txt 2>&1; cat $OUT/traffic_profile.txt
rm -rf gpurun_out/prof_rows
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_rows -- python3 scripts/rows_workload.py > $OUT/rows_workload.txt 2>&1
timeout -k 10 120 python scripts/saturation_timing.py 1 2 4 8 > $OUT/saturation.txt 2>&1
PANTR_LIB_PATH=$P/lib/libpantr_mi355x_phase.so timeout -k 10 300 python scripts/phase_timing.py > $OUT/phase_timing.txt 2>&1
PANTR_LIB_PATH=$P/lib/libpantr_mi355x_phase.so timeout -k 10 300 python scripts/product_time_under_load.py > $OUT/product_under_load.txt 2>&1
for W in sweep stress; do
  PANTR_BENCH_REHEARSAL=1 timeout -k 10 300 python bench.py --gpus 2 --steps 3 --warmup 1 --workload $W --scaling strong > $OUT/rehearsal_strong_$W.log 2>&1
done
PANTR_BENCH_REHEARSAL=1 timeout -k 10 300 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/rehearsal_weak.log 2>&1
timeout -k 10 200 python bench.py --steps 10 --warmup 2 --streams 4 --headline-only > $OUT/bench_streams4.json 2> $OUT/bench_streams4.err
echo done > $OUT/done
