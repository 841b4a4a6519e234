#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out
for wl in c2 c3 c5; do
  timeout -k 10 400 python bench.py --workload $wl --no-cpu-baseline 2>&1 | grep '"metric"' | tee gpurun_out/bench_$wl.json | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['workload'][:40], d['value'], d['ms_per_step'], d.get('replay_stats'))" || exit 1
done
