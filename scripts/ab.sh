#!/usr/bin/env bash
# A/B of library variants built from ONE tree, on ONE box (run through gpurun): for each library given, optional
# parity tests, then the chosen timings, ROUNDS times round-robin so that drift of the box shows.  Replaces the
# one-off batches of earlier rounds.  Variants are built on the CPU box with
#     make -C pantr-cdc2023-experiments_amd LIB=$PWD/pantr-cdc2023-experiments_amd/lib/libpantr_<tag>.so EXTRA_HIPFLAGS=-D...
#     make -C pantr-cdc2023-experiments_amd phase PHASE_LIB=.../lib/libpantr_<tag>_phase.so EXTRA_HIPFLAGS=...
# (a stale variant of an older tree needs PANTR_ALLOW_STALE_LIB=1).
#
#   scripts/ab.sh [-r ROUNDS] [-t "<pytest arguments>"] [-w "<workloads>"] [-o <tag>] lib/<a>.so [lib/<b>.so ...]
#   workloads (default "headline"):
#     headline  pass time of the 600-instance bench batch and its halves   (scripts/gpu_timing.py)
#     tile8 | tile4 | stress   the memory-bound launches                   (scripts/traffic_workload.py)
#     hv        a lone product and 600 at once                             (scripts/hv_timing.py)
#     phase     per-phase means; needs a profiling build                   (scripts/phase_timing.py)
#     quad      pass times of the quadcopter-family configurations         (scripts/quad_configs_timing.py)
#     sweep     configs[3], six concurrent launches                        (scripts/sweep_speculation.py)
#     bench     the bench line                                             (bench.py --no-cpu-baseline)
#   results: gpurun_out/ab/<tag>.txt (and on stdout)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
P=$PWD/pantr-cdc2023-experiments_amd
ROUNDS=2; TESTS=""; WORK="headline"; TAG=ab
while getopts "r:t:w:o:" o; do
  case $o in r) ROUNDS=$OPTARG;; t) TESTS=$OPTARG;; w) WORK=$OPTARG;; o) TAG=$OPTARG;; *) exit 2;; esac
done
shift $((OPTIND - 1))
mkdir -p gpurun_out/ab; OUT=gpurun_out/ab/$TAG.txt; : > $OUT
filt() { grep -v "amdgpu.ids"; }
one_json() { grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['workload'], [round(x,1) for x in d['kernel_ms']], round(d['steps_per_s']), d['kernel'])"; }
if [ -n "$TESTS" ]; then
  for lib in "$@"; do
    echo "== tests: $lib" | tee -a $OUT
    PANTR_LIB_PATH=$P/$lib timeout -k 10 900 python -m pytest $TESTS -x -q -m gpu 2>&1 | tail -n 4 | tee -a $OUT
    [ ${PIPESTATUS[0]} -eq 0 ] || exit 1
  done
fi
for round in $(seq 1 $ROUNDS); do
  for lib in "$@"; do
    export PANTR_LIB_PATH=$P/$lib
    echo "== $lib (round $round)" | tee -a $OUT
    for w in $WORK; do
      case $w in
        headline) timeout -k 10 200 python scripts/gpu_timing.py 2>&1 | filt | grep "both600\|cold300\|warm300" | cut -c1-170;;
        tile*|stress) timeout -k 10 300 python scripts/traffic_workload.py $w 2>&1 | one_json;;
        hv) timeout -k 10 120 python scripts/hv_timing.py 1 2>&1 | filt; timeout -k 10 120 python scripts/hv_timing.py 600 2>&1 | filt;;
        phase) timeout -k 10 300 python scripts/phase_timing.py 2>&1 | filt;;
        quad) timeout -k 10 400 python scripts/quad_configs_timing.py 2>&1 | filt;;
        sweep) timeout -k 10 300 python scripts/sweep_speculation.py 2>&1 | filt;;
        bench) timeout -k 10 600 python bench.py --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('headline', round(d['value'], 1), 'steps/s', round(d['ms_per_step'], 2), 'ms; frac', d['roofline']['frac'])
for k, v in d.get('configs', {}).items():
    print(f\"  {k:22s} {v['value']:10.1f} steps/s  {v['ms_per_pass']:9.2f} ms  {v['roofline']['kernel']}\")";;
        *) echo "unknown workload $w";;
      esac
    done 2>&1 | tee -a $OUT
  done
done
echo finished
