#!/bin/bash
# Round-4 evidence on one box: the full GPU suite, bench lines (driver form, default form with the CPU baseline, c3, c5) and the
# kernel-trace / PMC profiles of the three graphed workloads.  Results under gpurun_out/final4/ (copied into profiles/round4/).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/final4; mkdir -p $out
export ISDQN_BUILD_TAG="round 4 HEAD (head kernels on the side queue under the dense data gradient, data-gradient chain first behind every fork; hardware rcp / sqrt in the Adam element; fused-Adam GEMM streams p / m / v nontemporally, weight-gradient slab stores likewise; mirror rebuilt at the head of every replay; 13x13 dz image in the conv2 data gradient; weight-gradient groups follow the batch; XCD-affine image order in the conv1 pair kernel)"
timeout -k 10 1000 python -m pytest tests -q -m gpu -x > $out/gputests_full_suite.log 2>&1 || { tail -20 $out/gputests_full_suite.log; exit 1; }
tail -2 $out/gputests_full_suite.log
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_c2_driver_form.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
timeout -k 10 400 python bench.py > $out/bench_c2.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
for wl in c3 c5; do
  timeout -k 10 300 python bench.py --workload $wl --no-cpu-baseline > $out/bench_$wl.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
done
python3 -c "
import json
for w in ('c2_driver_form','c2','c3','c5'):
    d=json.loads(open('$out/bench_%s.json'%w).read().strip().splitlines()[-1]); print(w, round(d['value'],1), 'steps/s', round(d['roofline']['frac'],4), 'of attainable', round(d['roofline']['frac_of_attainable'],3))
"
for wl in c2 c3 c5; do
  bash scripts/r3/profile.sh f4$wl $wl > $out/profile_$wl.log 2>&1 || { tail -5 $out/profile_$wl.log; exit 1; }
  mkdir -p $out/$wl; cp gpurun_out/prof_f4$wl/table.md $out/$wl/${wl}_kernel_roofline.md; cp gpurun_out/prof_f4$wl/timeline.txt $out/$wl/${wl}_timeline.txt
  cp gpurun_out/prof_f4$wl/${wl}_hbm_traffic.json $out/$wl/; cp gpurun_out/prof_f4$wl/kernel_stats.csv $out/$wl/${wl}_kernel_stats.csv 2>/dev/null
  cp gpurun_out/prof_f4$wl/bench_under_trace.json $out/$wl/${wl}_bench_under_trace.json
  tail -3 $out/profile_$wl.log | head -1
done
