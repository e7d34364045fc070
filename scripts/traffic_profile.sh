#!/usr/bin/env bash
# rocprofv3 passes for the memory-bound workloads (missing item 4): kernel trace + stats, then
# FETCH_SIZE and WRITE_SIZE in passes of their own (never combined with another trace domain), for
#   stress : BASELINE.json configs[4] on one GPU (N = 120, 1024 instances)
#   tile8  : the bench batch tiled 8 times (4800 instances)
#   realistic_quadcopter | quadcopter_N12 : BASELINE.json configs[2] | configs[0] (the replay batches bench.py times)
# and, SQ=1 in the environment, a fourth pass with the occupancy / issue counters (SQ_WAVES SQ_INSTS_VALU SQ_BUSY_CYCLES
# SQ_WAIT_ANY, then SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU: two passes of their own).
# The program stands directly behind `--` (no env / bash -c hop between rocprofv3 and python3).
# Output: gpurun_out/traffic/<workload>/{stats,pmc_fetch,pmc_write[,pmc_sq,pmc_sq2]}; digest: scripts/summarize_traffic.py <tag>
set -e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
for W in "$@"; do
    name=${W%% *}
    OUT=gpurun_out/traffic/${W// /_}
    rm -rf "$OUT"; mkdir -p "$OUT"
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 scripts/traffic_workload.py $W > $OUT/stats.log 2>&1
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 scripts/traffic_workload.py $W > $OUT/pmc_fetch.log 2>&1
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 scripts/traffic_workload.py $W > $OUT/pmc_write.log 2>&1
    if [ -n "$SQ" ]; then
        rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAIT_ANY --output-format csv -d $OUT/pmc_sq -- python3 scripts/traffic_workload.py $W > $OUT/pmc_sq.log 2>&1
        rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc_sq2 -- python3 scripts/traffic_workload.py $W > $OUT/pmc_sq2.log 2>&1
    fi
    grep -h '^{' $OUT/stats.log | tail -1
    echo "done $W"
done
