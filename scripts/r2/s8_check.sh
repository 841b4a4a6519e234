#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export PYTHONUNBUFFERED=1
bash scripts/r2/bits_ab.sh pre_s8 s8p4 || exit 1
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 || exit 1
bash scripts/ab_bench.sh is-dqn_amd/lib/libisdqn_hip_pre_s8.so is-dqn_amd/lib/libisdqn_hip.so 3
