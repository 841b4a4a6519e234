#!/bin/bash
# bench the development build under a list of environment settings: scripts/r2/sweep_env.sh <workload> "VAR=1 VAR2=3" "..." ...
cd "$GRAFT_REPO_ROOT"
export PYTHONUNBUFFERED=1 ISDQN_HIP_LIB=$PWD/is-dqn_amd/lib/libisdqn_hip_dev.so
wl=$1; shift
for cfg in "$@"; do
  v=$(env $cfg timeout -k 10 200 python bench.py --workload $wl --no-cpu-baseline --steps 2400 --warmup 800 --replay-stats 0 2>/dev/null | grep '"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f steps/s  %.4f ms' % (d['value'], d['ms_per_step']))") || exit 1
  echo "[$cfg]: $v"
done
