#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export PYTHONUNBUFFERED=1
timeout -k 10 900 python -m pytest tests/test_gpu_graphed_update.py tests/test_gpu_agent.py tests/test_gpu_replay_buffer.py -x -q 2>&1 | tail -6 &&
bash scripts/r2/timeline.sh ${1:-q3} c3
