#!/bin/bash
# SQ counters of the conv kernels (two --pmc passes of 8 counters): where the wave cycles of a kernel go
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
pat=${1:-conv_fwd}
for pass in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" \
            "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM"; do
  out=gpurun_out/sq_ws; rm -rf $out; mkdir -p $out
  timeout -k 10 400 rocprofv3 --pmc $pass --output-format csv -d $out/pmc -- python3 bench.py --workload ${WL:-c2} --no-cpu-baseline --steps 80 --warmup 40 --settle 0 --graph 8 --replay-stats 0 > $out/pmc.log 2>&1 || { tail -8 $out/pmc.log; exit 1; }
  python3 - $(find $out/pmc -name "*_counter_collection.csv" | head -1) "$pat" <<'PY'
import sys, re, pandas as pd
t = pd.read_csv(sys.argv[1]); t = t[t.Kernel_Name.str.contains(sys.argv[2])]
t["k"] = t.Kernel_Name.map(lambda n: re.sub(r"\(.*", "", n.replace("void isdqn::", ""))[:48])
p = t.pivot_table(index="k", columns="Counter_Name", values="Counter_Value", aggfunc="sum")
n = t.groupby(["k", "Counter_Name"]).size().groupby("k").max()
pd.set_option("display.width", 250)
print((p.div(n, axis=0)).round(0).to_string())
PY
  rm -rf $out
done
