#!/bin/bash
# one --pmc pass of SQ counters over bench.py's graphed workload (counters alone: no trace domains in the same run)
# usage: scripts/r2/sq_counters.sh <workload>
wl=${1:-c2}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/sq_$wl; rm -rf $out; mkdir -p $out
timeout -k 10 500 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $out/pmc -- python3 bench.py --workload $wl --no-cpu-baseline --steps 160 --warmup 80 --settle 0 --graph 8 --replay-stats 0 > $out/pmc.log 2>&1 || { tail -8 $out/pmc.log; exit 1; }
python3 scripts/r2/sq_table.py $(find $out/pmc -name "*_counter_collection.csv" | head -1) > $out/sq_table.md 2> $out/sq_table.err || { cat $out/sq_table.err; exit 1; }
cat $out/sq_table.md
find $out -name "*_counter_collection.csv" -size +20M -delete
