#!/bin/bash
# the driver's form (--steps 20 --warmup 5) on one box, alternating variants.  usage: driver_form_variants.sh "<flags A>" "<flags B>" ...
run() { timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --replay-stats 0 $1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), 'steps/s  device', round(d['roofline']['device_ms_per_step_avg'],4), 'ms  host issue', round(d['roofline']['host_issue_ms_per_step'],4), 'ms ', d['config']['launch'])"; }
for i in 1 2 3; do for v in "$@"; do echo -n "[$v] "; run "$v" || exit 1; done; done
