#!/bin/bash
# engine / memory clocks and power while bench.py's default form runs (read-only rocm-smi samples every 0.5 s)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/clocks.txt; : > $out
(timeout -k 5 120 python bench.py --no-cpu-baseline --steps 80000 --warmup 4000 --replay-stats 0 > gpurun_out/clocks_bench.json 2>/dev/null) &
bp=$!
sleep 9
for i in $(seq 1 30); do
  /opt/rocm/bin/rocm-smi -d 0 --showclocks --showpower --showperflevel 2>/dev/null | grep -E "sclk|mclk|fclk|Power|Performance" | tr '\n' ';' >> $out; echo >> $out
  sleep 0.5
  kill -0 $bp 2>/dev/null || break
done
wait $bp
echo "idle:" >> $out
sleep 2
/opt/rocm/bin/rocm-smi -d 0 --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" | tr '\n' ';' >> $out; echo >> $out
cat $out
python -c "import json; d=json.loads(open('gpurun_out/clocks_bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
