#!/usr/bin/env bash
# rocprofv3 passes for the headline bench (run on the GPU box through gpurun).
#   1. kernel trace + stats         -> gpurun_out/prof/stats
#   2. PMC pass: FETCH_SIZE          -> gpurun_out/prof/pmc_fetch
#   3. PMC pass: WRITE_SIZE          -> gpurun_out/prof/pmc_write
#   4. PMC pass: SQ busy/wait counters
#   5. PMC pass: LDS activity and bank conflicts
# Counters are collected in their own runs (never combined with other trace domains).
set -e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
OUT=gpurun_out/prof
rm -rf $OUT
mkdir -p $OUT
ARGS="bench.py --steps 3 --warmup 1 --headline-only"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ARGS > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ARGS > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU --output-format csv -d $OUT/pmc_sq -- python3 $ARGS > $OUT/pmc_sq.log 2>&1 || echo "sq pass failed" >> $OUT/pmc_sq.log
# LDS: how busy the CUs' local data stores are (the instances sharing a CU contend for them), bank conflicts
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS --output-format csv -d $OUT/pmc_lds -- python3 $ARGS > $OUT/pmc_lds.log 2>&1 || echo "lds pass failed" >> $OUT/pmc_lds.log
find $OUT -name "*.csv" | head -50
