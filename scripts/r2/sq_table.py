"""Per-kernel SQ counter table of one rocprofv3 --pmc pass (scripts/r2/sq_counters.sh): where the wave time goes.
WAIT_ANY = parked on s_waitcnt / barrier, WAIT_INST_ANY = issue stalls, ACTIVE_INST_ANY = issuing (the three are disjoint and
add up to about WAVE_CYCLES; all in quad-cycles), LDS bank-conflict share of the LDS array cycles, and the MFMA pipe's busy
cycles against the kernel's own duration x SIMDs.  usage: sq_table.py <counter_collection.csv> [<kernel_trace.csv>]"""
import re
import sys

import pandas as pd

t = pd.read_csv(sys.argv[1])
t["k"] = t.Kernel_Name.map(lambda n: re.sub(r"\(.*", "", re.sub(r"^void ", "", n).replace("isdqn::", ""))[:100])
t = t[~t.k.str.startswith("at::") & ~t.k.str.contains("rocclr")]
p = t.pivot_table(index="k", columns="Counter_Name", values="Counter_Value", aggfunc="sum")
n = t[t.Counter_Name == t.Counter_Name.iloc[0]].groupby("k").size()
cols = list(p.columns)
print("| kernel | launches | wave quad-cycles / launch | parked (WAIT_ANY) | issue stall (WAIT_INST_ANY) | issuing (ACTIVE_INST_ANY) | LDS conflict cycles / LDS cycles | MFMA busy cycles / launch |")
print("|---|---|---|---|---|---|---|---|")
for k, r in p.sort_values("SQ_WAVE_CYCLES", ascending=False).iterrows():
    wc = r.get("SQ_WAVE_CYCLES", float("nan"))
    if not wc or n[k] < 20:
        continue
    f = lambda c: r.get(c, float("nan")) / wc
    lds = r.get("SQ_LDS_BANK_CONFLICT", float("nan")) / max(r.get("SQ_LDS_IDX_ACTIVE", float("nan")), 1)
    print(f"| `{k}` | {n[k]} | {wc / n[k]:.3g} | {f('SQ_WAIT_ANY'):.2f} | {f('SQ_WAIT_INST_ANY'):.2f} | {f('SQ_ACTIVE_INST_ANY'):.2f} | {lds:.3f} | {r.get('SQ_VALU_MFMA_BUSY_CYCLES', float('nan')) / n[k]:.3g} |")
print("\ncounters:", ", ".join(cols))
