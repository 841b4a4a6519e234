#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export PYTHONUNBUFFERED=1
timeout -k 10 1000 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 || exit 1
bash scripts/ab_bench.sh is-dqn_amd/lib/libisdqn_hip.so is-dqn_amd/lib/libisdqn_hip_d64.so 2
