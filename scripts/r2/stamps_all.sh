#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export ISDQN_HIP_LIB=$PWD/is-dqn_amd/lib/libisdqn_hip_dev.so
for l in Conv_0 Conv_1 Conv_2 dgrad:Conv_2 dgrad:Conv_1 head_chain; do
  echo "=== $l"; GHZ=2.4 timeout -k 10 120 python scripts/stamps.py $l 2>&1 | grep -v "amdgpu.ids\|^(layer\|histogram" | head -22
done
