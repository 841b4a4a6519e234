#!/bin/bash
# timeline rows (grep pattern $2) of the development build under a list of environment settings
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export ISDQN_HIP_LIB=$PWD/is-dqn_amd/lib/libisdqn_hip_dev.so
wl=$1; pat=$2; shift 2
for cfg in "$@"; do
  out=gpurun_out/tl_sweep; rm -rf $out; mkdir -p $out
  env $cfg timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/kt -- python3 bench.py --workload $wl --no-cpu-baseline --steps 400 --warmup 200 --settle 0 --replay-stats 0 > $out/kt.log 2>&1 || { tail -5 $out/kt.log; exit 1; }
  echo "== [$cfg]"
  python3 scripts/r2/timeline.py $(find $out/kt -name "*_kernel_trace.csv" | head -1) | grep -E "period|$pat"
  rm -rf $out
done
