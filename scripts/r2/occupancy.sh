#!/bin/bash
# workgroups per CU of every hot kernel at a workload's shapes (development build)
cd "$GRAFT_REPO_ROOT"
export PYTHONUNBUFFERED=1 ISDQN_HIP_LIB=$PWD/is-dqn_amd/lib/libisdqn_hip_dev.so ISDQN_DEBUG_OCCUPANCY=1
mkdir -p gpurun_out
timeout -k 10 300 python bench.py --workload ${1:-c2} --no-cpu-baseline --steps 16 --warmup 8 --settle 0 --replay-stats 0 2> gpurun_out/occupancy_${1:-c2}.txt | cut -c1-150
grep "isdqn" gpurun_out/occupancy_${1:-c2}.txt | sed 's/static int isdqn:://; s/\[with //' | cut -c1-260
