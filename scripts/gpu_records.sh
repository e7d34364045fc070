#!/usr/bin/env bash
# The records of one round on the GPU box (run through gpurun), all written under gpurun_out/<tag>/.  Two parts, one gpurun
# call each (a call is limited to twenty minutes):
#   scripts/gpu_records.sh <tag> a   the GPU test suite, the bench line, rocprofv3 kernel trace + PMC passes of the headline
#                                    bench (scripts/profile.sh -> gpurun_out/prof), kernel trace + FETCH_SIZE / WRITE_SIZE
#                                    passes of the memory-bound workloads (scripts/traffic_profile.sh stress tile8)
#   scripts/gpu_records.sh <tag> b   the same passes + the SQ counters for the quadcopter configurations (configs[2], [0]),
#                                    the kernel trace of the other solve kernels (scripts/rows_workload.py), the batch tiled
#                                    R times, per-phase times (profiling build), 4800 genuine closed loops with and without the
#                                    start order of the step before, the multi-rank bench modes started bare
#                                    with two ranks on the one GPU (gloo, PANTR_BENCH_REHEARSAL=1), bench.py --streams 4
# Afterwards, on the build box: scripts/collect_records.sh <tag>, then (the PMC digests now match the sources) once more
#   python bench.py > gpurun_out/<tag>/bench.json   for the line whose roofline.traffic fields are filled.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
TAG=${1:-r05}; PART=${2:-a}
OUT=gpurun_out/$TAG
P=$PWD/pantr-cdc2023-experiments_amd
mkdir -p $OUT
if [ "$PART" = a ]; then
  timeout -k 10 900 python -m pytest tests -q -m gpu > $OUT/gpu_tests.log 2>&1; tail -2 $OUT/gpu_tests.log
  ( time timeout -k 10 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2> $OUT/bench_time.txt; python scripts/bench_summary.py $OUT/bench.json; cat $OUT/bench_time.txt
  timeout -k 10 400 bash scripts/profile.sh > $OUT/profile.log 2>&1; tail -3 $OUT/profile.log
  timeout -k 10 600 bash scripts/traffic_profile.sh stress tile8 > $OUT/traffic_profile.txt 2>&1; cat $OUT/traffic_profile.txt
else
  SQ=1 timeout -k 10 600 bash scripts/traffic_profile.sh realistic_quadcopter quadcopter_N12 > $OUT/traffic_profile_quad.txt 2>&1; cat $OUT/traffic_profile_quad.txt
  rm -rf gpurun_out/prof_rows
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_rows -- python3 scripts/rows_workload.py > $OUT/rows_workload.txt 2>&1
  timeout -k 10 120 python scripts/saturation_timing.py 1 2 4 8 > $OUT/saturation.txt 2>&1
  PANTR_LIB_PATH=$P/lib/libpantr_mi355x_phase.so timeout -k 10 300 python scripts/phase_timing.py > $OUT/phase_timing.txt 2>&1
  PANTR_LIB_PATH=$P/lib/libpantr_mi355x_phase.so timeout -k 10 200 python scripts/straggler_phases.py auto > $OUT/straggler_phases.txt 2>&1
  PANTR_LIB_PATH=$P/lib/libpantr_mi355x_phase.so timeout -k 10 300 python scripts/product_time_under_load.py launch_order=auto > $OUT/product_under_load.txt 2>&1
  timeout -k 10 200 python scripts/closed_loop_order.py 4800 16 > $OUT/closed_loop_order.txt 2>&1
  for W in sweep stress; do
    PANTR_BENCH_REHEARSAL=1 timeout -k 10 300 python bench.py --gpus 2 --steps 3 --warmup 1 --workload $W --scaling strong > $OUT/rehearsal_strong_$W.log 2>&1
  done
  PANTR_BENCH_REHEARSAL=1 timeout -k 10 400 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/rehearsal_weak.log 2>&1
  timeout -k 10 200 python bench.py --steps 10 --warmup 2 --streams 4 --headline-only > $OUT/bench_streams4.json 2> $OUT/bench_streams4.err
  grep -h "^{" $OUT/rehearsal_weak.log | tail -1 > $OUT/rehearsal_weak.json; python scripts/bench_summary.py $OUT/rehearsal_weak.json
fi
echo done > $OUT/done_$PART
