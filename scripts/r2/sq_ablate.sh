#!/bin/bash
# LDS bank-conflict share of the forward conv kernels with phases ablated (development build): which phase conflicts?
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export ISDQN_HIP_LIB=$PWD/is-dqn_amd/lib/libisdqn_hip_dev.so
for ab in 0 6 5 3; do
  out=gpurun_out/sq_ab$ab; rm -rf $out; mkdir -p $out
  ISDQN_ABLATE=$ab timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $out/pmc -- python3 bench.py --workload c2 --no-cpu-baseline --steps 80 --warmup 40 --settle 0 --graph 8 --replay-stats 0 > $out/pmc.log 2>&1 || { tail -8 $out/pmc.log; exit 1; }
  echo "== ISDQN_ABLATE=$ab (0 full; 6 fill only; 5 K loop only; 3 epilogue only)"
  python3 - $(find $out/pmc -name "*_counter_collection.csv" | head -1) <<'PY'
import sys, re, pandas as pd
t = pd.read_csv(sys.argv[1]); t = t[t.Kernel_Name.str.contains("conv_fwd_img")]
t["k"] = t.Kernel_Name.map(lambda n: re.sub(r"\(.*", "", n.replace("void isdqn::", "")))
p = t.pivot_table(index="k", columns="Counter_Name", values="Counter_Value", aggfunc="sum")
n = t[t.Counter_Name == "SQ_WAVE_CYCLES"].groupby("k").size()
for k, r in p.iterrows():
    print(f"  {k:42s} LDS cycles/launch {r.SQ_LDS_IDX_ACTIVE / n[k]:10.0f}  conflict cycles/launch {r.SQ_LDS_BANK_CONFLICT / n[k]:10.0f}  share {r.SQ_LDS_BANK_CONFLICT / max(r.SQ_LDS_IDX_ACTIVE, 1):.3f}")
PY
  rm -rf $out
done
