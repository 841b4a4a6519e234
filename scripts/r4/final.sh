#!/bin/bash
# Round-4 evidence on one box: the full GPU suite, bench lines (driver form, default form with the CPU baseline, c3, c5) and the
# kernel-trace / PMC profiles of the three graphed workloads.  Results under gpurun_out/final4/ (copied into profiles/round4/).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/final4; mkdir -p $out
export ISDQN_BUILD_TAG="round 4 HEAD (weight-gradient groups prefetch their next image into registers; head kernels on the side queue under the dense data gradient, data-gradient chain first behind every fork; hardware rcp / sqrt in the Adam element; fused-Adam GEMM streams p / m / v nontemporally, weight-gradient slab stores likewise; mirror rebuilt at the head of every replay; 13x13 dz image in the conv2 data gradient; weight-gradient groups follow the batch; XCD-affine image order in the conv1 pair kernel)"
timeout -k 10 1000 python -m pytest tests -q -m gpu -x > $out/gputests_full_suite.log 2>&1 || { tail -20 $out/gputests_full_suite.log; exit 1; }
tail -2 $out/gputests_full_suite.log
for wl in c2 c3 c5; do
  bash scripts/r3/profile.sh f4$wl $wl > $out/profile_$wl.log 2>&1 || { tail -5 $out/profile_$wl.log; exit 1; }
  mkdir -p $out/$wl; cp gpurun_out/prof_f4$wl/table.md $out/$wl/${wl}_kernel_roofline.md; cp gpurun_out/prof_f4$wl/timeline.txt $out/$wl/${wl}_timeline.txt
  cp gpurun_out/prof_f4$wl/${wl}_hbm_traffic.json $out/$wl/; cp gpurun_out/prof_f4$wl/kernel_stats.csv $out/$wl/${wl}_kernel_stats.csv 2>/dev/null
  cp gpurun_out/prof_f4$wl/bench_under_trace.json $out/$wl/${wl}_bench_under_trace.json
  cp gpurun_out/prof_f4$wl/${wl}_hbm_traffic.json profiles/round4/${wl}_hbm_traffic_v10.json  # (on the box: the bench lines below quote it)
  tail -3 $out/profile_$wl.log | head -1
done
bash scripts/r4/bench_lines.sh || exit 1
