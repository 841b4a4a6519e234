#!/bin/bash
# timeline of the product build and of the development build side by side (same box)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for tag in product dev; do
  if [ $tag = dev ]; then export ISDQN_HIP_LIB=$PWD/is-dqn_amd/lib/libisdqn_hip_dev.so; fi
  out=gpurun_out/tl_ab_$tag; rm -rf $out; mkdir -p $out
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/kt -- python3 bench.py --workload ${1:-c2} --no-cpu-baseline --steps 800 --warmup 400 --replay-stats 0 > $out/kt.log 2>&1 || { tail -5 $out/kt.log; exit 1; }
  echo "== $tag"; python3 scripts/r2/timeline.py $(find $out/kt -name "*_kernel_trace.csv" | head -1)
  rm -rf $out/kt
done
