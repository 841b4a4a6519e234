#!/bin/bash
# how much of the first layer's kernels is the latency of cold replay frames?  c2 with a 1e6-element store (7 GB) against a 4096-element one (29 MB: cache resident)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for cap in 1000000 4096; do
  out=gpurun_out/hot_$cap; rm -rf $out; mkdir -p $out
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/kt -- python3 bench.py --workload c2 --no-cpu-baseline --steps 800 --warmup 400 --replay-stats 0 --capacity $cap > $out/kt.log 2>&1 || { tail -5 $out/kt.log; exit 1; }
  echo "capacity $cap: $(grep -o '"value": [0-9.]*' $out/kt.log | head -1)"
  python3 scripts/r2/timeline.py $(find $out/kt -name "*_kernel_trace.csv" | head -1) | grep -E "period|u8|wgrad_img_kernel<2"
  rm -rf $out/kt
done
