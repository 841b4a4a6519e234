#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export PYTHONUNBUFFERED=1
export ISDQN_HIP_LIB=$PWD/is-dqn_amd/lib/libisdqn_hip_bi_n1d64_noslp.so
echo "== candidate (builtin MFMA, one accumulator set, 64-row DenseDgradLN, no SLP): hunt c2 x3000, hunt2, c5 hunt"
REPS=3000 timeout -k 10 400 python scripts/dbg_hunt.py 2>&1 | grep -v amdgpu.ids | tail -6 || exit 1
REPS=300 STEPS=4 timeout -k 10 300 python scripts/dbg_hunt2.py 2>&1 | grep -v amdgpu.ids | tail -4 || exit 1
B=1024 K=32 A=4 REPS=300 timeout -k 10 400 python scripts/dbg_hunt.py 2>&1 | grep -v amdgpu.ids | tail -6 || exit 1
unset ISDQN_HIP_LIB
bash scripts/ab_bench.sh is-dqn_amd/lib/libisdqn_hip.so is-dqn_amd/lib/libisdqn_hip_bi_n1d64_noslp.so 2
