#!/bin/bash
# SQ instruction counters of bench.py's graphed workload, per kernel and per step (one --pmc pass, counters alone).
# usage: scripts/r4/sq_instructions.sh <workload>
wl=${1:-c2}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/sqi_$wl; rm -rf $out; mkdir -p $out
timeout -k 10 500 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $out/pmc -- python3 bench.py --workload $wl --no-cpu-baseline --steps 160 --warmup 80 --settle 0 --graph 8 --replay-stats 0 > $out/pmc.log 2>&1 || { tail -8 $out/pmc.log; exit 1; }
python3 - "$(find $out/pmc -name '*_counter_collection.csv' | head -1)" <<'PY' | tee gpurun_out/sq_instruction_counters_$wl.txt
import re, sys
import pandas as pd
df = pd.read_csv(sys.argv[1])
df["k"] = df.Kernel_Name.map(lambda n: re.sub(r"\(.*", "", re.sub(r"^void ", "", n).replace("isdqn::", ""))[:64])
steps = 160 + 80
t = df.pivot_table(index="k", columns="Counter_Name", values="Counter_Value", aggfunc="sum") / steps
t = t[~t.index.str.contains("at::native|rocclr|gather_rows|split_params")]
t = t.sort_values("SQ_INSTS_VALU", ascending=False)
pd.set_option("display.width", 250); pd.set_option("display.float_format", lambda v: f"{v:,.0f}")
print("# per step (240 steps of the run), wave-instructions / cycles as the counters report them")
print(t.to_string())
tot = t.sum()
print("\nper step: VALU %.1f M  SALU %.1f M  LDS %.2f M  VMEM read %.2f M  VMEM write %.2f M;  MFMA busy cycles %.1f M (= %.1f M MFMAs of 16 cycles)" % (
    tot.SQ_INSTS_VALU / 1e6, tot.SQ_INSTS_SALU / 1e6, tot.SQ_INSTS_LDS / 1e6, tot.SQ_INSTS_VMEM_RD / 1e6, tot.SQ_INSTS_VMEM_WR / 1e6,
    tot.SQ_VALU_MFMA_BUSY_CYCLES / 1e6, tot.SQ_VALU_MFMA_BUSY_CYCLES / 16e6))
PY
find $out -name "*_counter_collection.csv" -size +20M -delete
