#!/bin/bash
# quick check of a kernel-side change: network / graph / baseline parity tests, then the c2 timeline
cd "$GRAFT_REPO_ROOT"
export PYTHONUNBUFFERED=1
timeout -k 10 900 python -m pytest tests/test_gpu_network.py tests/test_gpu_graphed_update.py tests/test_gpu_dqn_baselines.py tests/test_gpu_fullsize_properties.py -x -q 2>&1 | tail -6 &&
bash scripts/r2/timeline.sh ${1:-q} c2
