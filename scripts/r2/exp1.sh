#!/bin/bash
# round-2 experiment 1: asm MFMA (product) vs builtin MFMA: determinism hunt + bench
cd "$GRAFT_REPO_ROOT"
export PYTHONUNBUFFERED=1
for v in "" _builtin; do
  export ISDQN_HIP_LIB=$PWD/is-dqn_amd/lib/libisdqn_hip$v.so
  echo "== lib$v hunt c2" 
  REPS=1500 timeout -k 10 300 python scripts/dbg_hunt.py 2>&1 | tail -15 || exit 1
  echo "== lib$v hunt2"
  REPS=200 STEPS=4 timeout -k 10 200 python scripts/dbg_hunt2.py 2>&1 | tail -8 || exit 1
done
bash scripts/ab_bench.sh is-dqn_amd/lib/libisdqn_hip.so is-dqn_amd/lib/libisdqn_hip_builtin.so 2
