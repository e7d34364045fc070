#!/usr/bin/env bash
# Full GPU test suite, then the bench line with a short summary of its configs (development aid).
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/r04/check; mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; rc=$?
tail -n 3 $O/gpu_tests.log
[ $rc -eq 0 ] || exit $rc
python bench.py "$@" > $O/bench.json 2> $O/bench.err || { tail -c 1500 $O/bench.err; exit 1; }
python - <<PY
import json
d = json.load(open("$O/bench.json"))
print("headline", round(d["value"], 1), "steps/s", round(d["ms_per_step"], 2), "ms; roofline frac", d["roofline"]["frac"])
for k, v in d.get("configs", {}).items():
    print(f"  {k:22s} {v['value']:10.1f} steps/s  {v['ms_per_pass']:9.2f} ms  {v['roofline']['kernel']}")
PY
