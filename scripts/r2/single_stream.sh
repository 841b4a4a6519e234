#!/bin/bash
# isolated kernel times: the dev build with the weight-gradient side stream disabled (every kernel alone on the chip)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export ISDQN_HIP_LIB=$PWD/is-dqn_amd/lib/libisdqn_hip_dev.so ISDQN_SINGLE_STREAM=1
timeout -k 10 280 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ss -- python3 bench.py --workload ${1:-c2} --no-cpu-baseline --steps 800 --warmup 400 --replay-stats 0 > gpurun_out/ss.log 2>&1 || exit 1
grep '"metric"' gpurun_out/ss.log | cut -c1-200
python scripts/kernel_means.py gpurun_out/ss
rm -rf gpurun_out/ss
