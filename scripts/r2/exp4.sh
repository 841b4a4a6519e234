#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export PYTHONUNBUFFERED=1
echo "== packed-fp32 hazard probe"
timeout -k 10 120 ./scripts/probes/pk_hazard_probe 4096 20000 || exit 1
echo "== hunt: builtin MFMA + NACC1 + DGRAD64 + -fno-slp-vectorize"
export ISDQN_HIP_LIB=$PWD/is-dqn_amd/lib/libisdqn_hip_bi_n1d64_noslp.so
REPS=1200 timeout -k 10 300 python scripts/dbg_hunt.py 2>&1 | grep -v amdgpu.ids | tail -12
