#!/bin/bash
# Round-3 profile of bench.py's own (graphed) workload: kernel trace + stats + steady-state timeline, then FETCH_SIZE / WRITE_SIZE
# in separate passes (MI355X_MICROARCH.md: one counter per pass, program directly behind `--`).
# usage: scripts/r3/profile.sh <tag> <workload: c2|c3|c5> [graph steps for the PMC passes, default 8]
tag=$1; wl=${2:-c2}; pg=${3:-8}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_$tag
mkdir -p $out
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 bench.py --workload $wl --no-cpu-baseline --steps 1600 --warmup 800 --replay-stats 0 > $out/kt.log 2>&1 || { tail -5 $out/kt.log; exit 1; }
grep '"metric"' $out/kt.log > $out/bench_under_trace.json
python3 scripts/r2/timeline.py $(find $out/kt -name "*_kernel_trace.csv" | head -1) > $out/timeline.txt 2>$out/timeline.err || cat $out/timeline.err
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $c --output-format csv -d $out/pmc_$c -- python3 bench.py --workload $wl --no-cpu-baseline --steps 160 --warmup 80 --settle 0 --graph $pg --replay-stats 0 > $out/pmc_$c.log 2>&1 || { tail -5 $out/pmc_$c.log; exit 1; }
done
python3 scripts/r2/kernel_table.py $out $wl 160 80 > $out/table.md 2>$out/table.err || { cat $out/table.err; exit 1; }
cat $out/table.md
cp $(find $out/kt -name "*_kernel_stats.csv" | head -1) $out/kernel_stats.csv 2>/dev/null
find $out -name "*_kernel_trace.csv" -size +20M -delete
find $out -name "*_counter_collection.csv" -size +20M -delete
