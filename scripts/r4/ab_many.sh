#!/bin/bash
# bench-only comparison of several builds, alternating.  usage: ab_many.sh <workload> <rounds> <variant or "-" for the product> ...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
wl=$1; n=$2; shift 2
for r in $(seq $n); do for v in "$@"; do
  if [ "$v" = "-" ]; then l=$PWD/is-dqn_amd/lib/libisdqn_hip.so; else l=$PWD/is-dqn_amd/lib/libisdqn_hip_$v.so; fi
  x=$(ISDQN_HIP_LIB=$l timeout -k 10 240 python bench.py --workload $wl --no-cpu-baseline --steps 2400 --warmup 800 --replay-stats 0 2>/dev/null | grep '"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f steps/s  %.4f ms' % (d['value'], d['ms_per_step']))") || exit 1
  echo "round $r [$v] $wl: $x"
done; done
