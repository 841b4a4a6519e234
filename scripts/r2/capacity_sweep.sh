#!/bin/bash
# does the random frame gather (TLB / DRAM page misses over a 7 GB pool) cost the first conv layer? timeline at small replay capacities
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for cap in 4096 65536 1000000; do
  out=gpurun_out/tl_cap$cap; mkdir -p $out
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/kt -- python3 bench.py --workload c2 --capacity $cap --no-cpu-baseline --steps 800 --warmup 400 --replay-stats 0 > $out/kt.log 2>&1 || { tail -5 $out/kt.log; exit 1; }
  echo "== capacity $cap"; grep '"metric"' $out/kt.log | cut -c60-130
  python3 scripts/r2/timeline.py $(find $out/kt -name "*_kernel_trace.csv" | head -1) | grep -E "period|conv_fwd_img_kernel<2|conv_wgrad_img_kernel<2"
  rm -rf $out/kt
done
