#!/bin/bash
# usage: scripts/r2/bits_ab.sh <variantA or ""> <variantB>: identical fingerprints?
cd "$GRAFT_REPO_ROOT"
a=is-dqn_amd/lib/libisdqn_hip${1:+_$1}.so; b=is-dqn_amd/lib/libisdqn_hip_$2.so
ISDQN_HIP_LIB=$PWD/$a timeout -k 10 300 python scripts/r2/bits.py 2>&1 | grep -v amdgpu.ids > gpurun_out/bits_A.txt || { tail -5 gpurun_out/bits_A.txt; exit 1; }
ISDQN_HIP_LIB=$PWD/$b timeout -k 10 300 python scripts/r2/bits.py 2>&1 | grep -v amdgpu.ids > gpurun_out/bits_B.txt || { tail -5 gpurun_out/bits_B.txt; exit 1; }
if diff gpurun_out/bits_A.txt gpurun_out/bits_B.txt > gpurun_out/bits_diff.txt; then echo "BIT-IDENTICAL ($(wc -l < gpurun_out/bits_A.txt) cases)"; else echo "DIFFERENT:"; cat gpurun_out/bits_diff.txt | cut -c1-400; fi
