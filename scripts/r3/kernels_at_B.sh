#!/bin/bash
# per-kernel mean durations of the captured step at a given batch size (kernel trace of bench.py --B)
B=${1:-32}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 280 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt_B$B -- python3 bench.py --B $B --capacity 200000 --no-cpu-baseline --steps 800 --warmup 400 --replay-stats 0 > gpurun_out/kt_B$B.log 2>&1 || { tail -5 gpurun_out/kt_B$B.log; exit 1; }
python3 scripts/r2/timeline.py $(find gpurun_out/kt_B$B -name "*_kernel_trace.csv" | head -1)
rm -rf gpurun_out/kt_B$B
