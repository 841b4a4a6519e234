#!/bin/bash
# steady-state timeline of bench.py's graphed step with a given library build.  usage: timeline_of.sh <variant or ""> <workload>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
v=$1; wl=${2:-c2}
[ -n "$v" ] && export ISDQN_HIP_LIB=$PWD/is-dqn_amd/lib/libisdqn_hip_$v.so
out=gpurun_out/tl_${v:-product}_$wl; rm -rf $out; mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/kt -- python3 bench.py --workload $wl --no-cpu-baseline --steps 800 --warmup 400 --replay-stats 0 > $out/kt.log 2>&1 || { tail -5 $out/kt.log; exit 1; }
python3 scripts/r2/timeline.py $(find $out/kt -name "*_kernel_trace.csv" | head -1) | tee gpurun_out/timeline_${v:-product}_$wl.txt
rm -rf $out/kt
