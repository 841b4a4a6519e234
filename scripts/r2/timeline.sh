#!/bin/bash
# usage: timeline.sh <tag> <workload> [extra bench args]
tag=$1; wl=${2:-c2}; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/tl_$tag
mkdir -p $out
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $out/kt -- python3 bench.py --workload $wl --no-cpu-baseline --steps 800 --warmup 400 --replay-stats 0 "$@" > $out/kt.log 2>&1 || { tail -5 $out/kt.log; exit 1; }
grep '"metric"' $out/kt.log | cut -c1-200
python3 scripts/r2/timeline.py $(find $out/kt -name "*_kernel_trace.csv" | head -1) > $out/timeline.txt 2>$out/timeline.err || { cat $out/timeline.err; exit 1; }
cat $out/timeline.txt
find $out -name "*_kernel_trace.csv" -size +20M -delete
