#!/bin/bash
# determinism hunt of the product build at HEAD: identical single steps from identical state must be bit-identical
cd "$GRAFT_REPO_ROOT"
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out
{
echo "== c2 shape x3000"; REPS=3000 timeout -k 10 500 python scripts/dbg_hunt.py 2>&1 | grep -v amdgpu.ids | tail -4 || exit 1
echo "== multi-step x300"; REPS=300 STEPS=4 timeout -k 10 300 python scripts/dbg_hunt2.py 2>&1 | grep -v amdgpu.ids | tail -3 || exit 1
echo "== c5 shape x300"; B=1024 K=32 A=4 REPS=300 timeout -k 10 400 python scripts/dbg_hunt.py 2>&1 | grep -v amdgpu.ids | tail -4 || exit 1
} | tee gpurun_out/hunt_final.log
