#!/bin/bash
# Round-end evidence on one box: the full GPU suite, the three bench lines, the profile of the graphed c2 workload.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/final
timeout -k 10 900 python -m pytest tests -q -m gpu -x > gpurun_out/final/gputests_full_suite.log 2>&1 || { tail -20 gpurun_out/final/gputests_full_suite.log; exit 1; }
tail -2 gpurun_out/final/gputests_full_suite.log
timeout -k 10 300 python bench.py > gpurun_out/final/bench_c2.json 2> gpurun_out/final/bench_c2.err || { tail -5 gpurun_out/final/bench_c2.err; exit 1; }
for wl in c3 c5; do
  timeout -k 10 300 python bench.py --workload $wl --no-cpu-baseline > gpurun_out/final/bench_$wl.json 2> gpurun_out/final/bench_$wl.err || { tail -5 gpurun_out/final/bench_$wl.err; exit 1; }
done
python3 -c "
import json
for w in ('c2','c3','c5'):
    d=json.loads(open('gpurun_out/final/bench_%s.json'%w).read().strip().splitlines()[-1]); print(w, round(d['value'],1), 'steps/s', round(d['roofline']['frac'],4))
"
bash scripts/r3/profile.sh r3final c2 > gpurun_out/final/profile.log 2>&1 || { tail -5 gpurun_out/final/profile.log; exit 1; }
tail -4 gpurun_out/final/profile.log
