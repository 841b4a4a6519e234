#!/bin/bash
# timeline of the product build and of a variant build side by side (same box): scripts/r2/ab_tl2.sh <variant> [workload]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for tag in product $1; do
  if [ $tag != product ]; then export ISDQN_HIP_LIB=$PWD/is-dqn_amd/lib/libisdqn_hip_$1.so; fi
  out=gpurun_out/tl_ab_$tag; rm -rf $out; mkdir -p $out
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/kt -- python3 bench.py --workload ${2:-c2} --no-cpu-baseline --steps 800 --warmup 400 --replay-stats 0 > $out/kt.log 2>&1 || { tail -5 $out/kt.log; exit 1; }
  echo "== $tag"; python3 scripts/r2/timeline.py $(find $out/kt -name "*_kernel_trace.csv" | head -1)
  rm -rf $out/kt
done
