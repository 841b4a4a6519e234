#!/bin/bash
# Independent replicas sharing one GPU: whole-chip streams (time-sliced) against CU-partitioned streams (side by side).
# usage: scripts/r4/replica_partition.sh [workload]   -> gpurun_out/r4_partition.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
wl=${1:-c2}; out=gpurun_out/r4_partition_$wl.txt; : > $out
run() {  # R graph partition-flag
  local tag="R=$1 graph=$2 ${3:-whole-chip}"
  v=$(timeout -k 10 280 python bench.py --workload $wl --capacity 200000 --no-cpu-baseline --replay-stats 0 --steps 800 --warmup 400 --replicas-per-gpu $1 --graph $2 $3 2>gpurun_out/r4_partition.err | grep '"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f steps/s aggregate  %.1f us per replica step  host issue %.1f us' % (d['value'], d['ms_per_step']*1e3, d['roofline']['host_issue_ms_per_step']*1e3))") || { echo "$tag FAILED: $(tail -2 gpurun_out/r4_partition.err)" | tee -a $out; return 0; }
  echo "$tag: $v" | tee -a $out
}
run 1 32
run 1 0
for R in 2 4 8; do
  run $R 0
  run $R 0 --cu-partition
  run $R 8
  run $R 8 --cu-partition
done
