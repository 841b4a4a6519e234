#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export PYTHONUNBUFFERED=1
export ISDQN_HIP_LIB=$PWD/is-dqn_amd/lib/libisdqn_hip_bi_n1d64.so
timeout -k 10 200 python scripts/r2/diag_part2.py 2>&1 | grep -v amdgpu.ids
