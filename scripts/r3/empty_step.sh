#!/bin/bash
# The step with every kernel returning at once (-DISDQN_EMPTY build): what its launches, graph edges and kernel boundaries cost by themselves.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
lib=$PWD/is-dqn_amd/lib/libisdqn_hip_empty.so
for B in 32 256; do
  ISDQN_HIP_LIB=$lib timeout -k 10 200 python bench.py --B $B --capacity 200000 --no-cpu-baseline --replay-stats 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('empty kernels, B', $B, round(d['ms_per_step']*1e3,1), 'us/step', d['config']['launch'])"
done
ISDQN_HIP_LIB=$lib timeout -k 10 200 python bench.py --graph 0 --steps 2000 --warmup 1000 --capacity 200000 --no-cpu-baseline --replay-stats 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('empty kernels, eager', round(d['ms_per_step']*1e3,1), 'us/step')"
ISDQN_HIP_LIB=$lib timeout -k 10 280 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/empty_kt -- python3 bench.py --capacity 200000 --no-cpu-baseline --steps 800 --warmup 400 --replay-stats 0 > gpurun_out/empty_kt.log 2>&1 || { tail -5 gpurun_out/empty_kt.log; exit 1; }
python3 scripts/r2/timeline.py $(find gpurun_out/empty_kt -name "*_kernel_trace.csv" | head -1)
rm -rf gpurun_out/empty_kt
