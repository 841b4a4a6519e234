#!/bin/bash
# round-2 experiment 2: the two formerly unstable kernels (single accumulator set, 64-row DenseDgradLN), asm vs builtin MFMA
cd "$GRAFT_REPO_ROOT"
export PYTHONUNBUFFERED=1
for v in _asm_n1d64 _bi_n1d64; do
  export ISDQN_HIP_LIB=$PWD/is-dqn_amd/lib/libisdqn_hip$v.so
  echo "== lib$v hunt c2" 
  REPS=1200 timeout -k 10 300 python scripts/dbg_hunt.py 2>&1 | tail -25 || exit 1
done
