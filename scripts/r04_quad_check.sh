#!/usr/bin/env bash
# Quadcopter-family GPU tests, then the pass times of its configurations (development aid).
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
python -m pytest tests/test_gpu_quadcopter.py tests/test_frozen_reference.py tests/test_gpu_panoc.py -x -q -m gpu 2>&1 | tail -n 3 || exit 1
timeout -k 10 400 python scripts/quad_configs_timing.py 2>&1 | grep -v amdgpu.ids
