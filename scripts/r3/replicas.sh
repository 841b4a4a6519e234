#!/bin/bash
# Independent replicas sharing ONE GPU on their own streams (the reference shares a GPU between seeds: launch_job/atari/normal/train.sh:9-16).
# usage: scripts/r3/replicas.sh <workload> <R...>
wl=${1:-c2}; shift
mkdir -p gpurun_out/replicas
for R in "$@"; do
  timeout -k 10 300 python3 bench.py --workload $wl --no-cpu-baseline --steps 2000 --warmup 2000 --replay-stats 0 --replicas-per-gpu $R > gpurun_out/replicas/${wl}_R$R.json 2> gpurun_out/replicas/${wl}_R$R.err || { tail -5 gpurun_out/replicas/${wl}_R$R.err; exit 1; }
  python3 -c "import json,sys; d=json.loads(open('gpurun_out/replicas/${wl}_R$R.json').read().strip().splitlines()[-1]); print('R=$R', round(d['value'],1), 'steps/s', round(d['ms_per_step'],4), 'ms per round of R steps')"
done
