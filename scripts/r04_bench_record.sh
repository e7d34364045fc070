#!/usr/bin/env bash
# The bench line once more after scripts/collect_records.sh wrote the PMC digests of the final sources (bench.py prints
# `roofline.traffic` only from digests whose source hash matches the tree's): gpurun_out/<tag>/bench.json
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
TAG=${1:-r04}; mkdir -p gpurun_out/$TAG
timeout -k 10 600 python bench.py > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err || { tail -c 1000 gpurun_out/$TAG/bench.err; exit 1; }
python - <<PY
import json
d = json.load(open("gpurun_out/$TAG/bench.json"))
print("headline", round(d["value"], 1), d["ms_per_step"], d["roofline"])
for k, v in d["configs"].items():
    print(k, round(v["value"], 1), round(v["ms_per_pass"], 2), v["roofline"].get("frac"), v["roofline"].get("traffic"))
print(d["cpu_baseline"])
PY
