#!/bin/bash
# the four bench lines of profiles/round4 (driver form, default form with the CPU baseline, c3, c5) with the current bench.py
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/final4; mkdir -p $out
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
