#!/bin/bash
# A/B two builds of libisdqn_hip.so on plain bench.py runs (no profiler), alternating, inside one GPU box.
# usage: scripts/ab_bench.sh <libA> <libB> [rounds]
cd "$GRAFT_REPO_ROOT"
for r in $(seq 1 ${3:-3}); do
  for tag in A B; do
    if [ $tag = A ]; then lib=$1; else lib=$2; fi
    v=$(ISDQN_HIP_LIB=$PWD/$lib timeout -k 10 200 python bench.py --no-cpu-baseline --steps 2000 --warmup 200 2>/dev/null | grep '"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f steps/s  %.4f ms' % (d['value'], d['ms_per_step']))") || exit 1
    echo "round $r $tag: $v"
  done
done
