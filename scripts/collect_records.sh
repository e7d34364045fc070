#!/usr/bin/env bash
# After scripts/gpu_records.sh <tag> ran on the GPU box (gpurun merges its output into gpurun_out/):
# copies what a round commits into profiles/<tag>_* and digests the profile passes.
#   scripts/collect_records.sh r03
set -e
cd "$(dirname "$0")/.."
TAG=${1:-r05}
R=gpurun_out/$TAG
for f in bench bench_streams4; do
    [ -s $R/$f.json ] && tail -1 $R/$f.json > profiles/${TAG}_$f.json
done
[ -s $R/gpu_tests.log ] && tail -3 $R/gpu_tests.log > profiles/${TAG}_gpu_tests.log
[ -s $R/phase_timing.txt ] && grep -v amdgpu.ids $R/phase_timing.txt > profiles/${TAG}_phase_timing.txt
for w in strong_stress strong_sweep weak; do
    grep '^{' $R/rehearsal_$w.log | tail -1 > profiles/${TAG}_rehearsal_2ranks_$w.json
done
grep -v amdgpu.ids $R/saturation.txt > profiles/${TAG}_saturation.txt
grep -v 'amdgpu.ids\|rocprofv3\|Opened result\|HSA version' $R/rows_workload.txt > profiles/${TAG}_rows_f1_f4_workload.txt
cp "$(ls -t gpurun_out/prof_rows/*/*_kernel_stats.csv | head -1)" profiles/${TAG}_rows_f1_f4_kernel_stats.csv
[ -s $R/product_under_load.txt ] && grep -v amdgpu.ids $R/product_under_load.txt > profiles/${TAG}_product_under_load.txt
python scripts/summarize_profiles.py $TAG > /dev/null
python scripts/summarize_traffic.py $TAG stress tile8
[ -d gpurun_out/traffic/realistic_quadcopter ] && python scripts/summarize_traffic.py $TAG realistic_quadcopter quadcopter_N12
# both quadcopter digests side by side in one file (configs[2] and configs[0])
python - $TAG <<'PY'
import json, os, sys
tag = sys.argv[1]
out = {}
for w in ("realistic_quadcopter", "quadcopter_N12"):
    f = f"profiles/{tag}_{w}_traffic.json"
    if os.path.exists(f):
        d = json.load(open(f))
        out[w] = {k: d[k] for k in ("command", "kernel_trace", "hbm_traffic_bytes_per_launch", "traffic_over_algorithmic",
                                     "valu_issue") if k in d}
        out[w]["csrc_sha256"] = d["run"].get("source_hash")
        out[w]["digest"] = f
if out:
    json.dump(out, open(f"profiles/{tag}_quad_summary.json", "w"), indent=1)
PY
[ -s $R/closed_loop_order.txt ] && grep -v amdgpu.ids $R/closed_loop_order.txt > profiles/${TAG}_closed_loop_order.txt
[ -s $R/straggler_phases.txt ] && grep -v amdgpu.ids $R/straggler_phases.txt > profiles/${TAG}_straggler_phases.txt
python scripts/kernel_resources.py > profiles/${TAG}_kernel_resources.txt 2> /dev/null || true
ls -la profiles/ | grep ${TAG}_
