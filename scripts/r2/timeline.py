"""Timeline of the steady-state step from a rocprofv3 kernel trace: for every kernel of the step its median start offset from
the step's first kernel, median duration, queue, and the gap to the end of the latest kernel that finished before it started
on the same queue.  usage: timeline.py <kernel_trace.csv>"""
import re
import sys

import numpy as np
import pandas as pd

df = pd.read_csv(sys.argv[1]).sort_values("Start_Timestamp").reset_index(drop=True)
df["k"] = df.Kernel_Name.map(lambda n: re.sub(r"\(.*", "", re.sub(r"^void ", "", n).replace("isdqn::", ""))[:90])
df = df.iloc[len(df) // 3:].reset_index(drop=True)
is_anchor = df.k.str.contains("conv_fwd_img_kernel<2, 2, true|conv_fwd_u8_pair_kernel<2") | df.k.str.contains("conv_fwd_img_kernel<2, 1, true|conv_fwd_u8_pair_kernel<1")
starts = np.flatnonzero(is_anchor.values)
rows = {}
periods = []
for a, b in zip(starts[:-1], starts[1:]):
    step = df.iloc[a:b]
    if len(step) > 40:
        continue
    t0 = step.Start_Timestamp.iloc[0]
    periods.append((df.Start_Timestamp.iloc[b] - t0) / 1e3)
    seen = {}
    for _, r in step.iterrows():
        n = seen.get(r.k, 0)
        seen[r.k] = n + 1
        rows.setdefault((r.k, n), []).append(((r.Start_Timestamp - t0) / 1e3, (r.End_Timestamp - r.Start_Timestamp) / 1e3, r.Queue_Id))
print(f"steps {len(periods)}  period median {np.median(periods):.1f} us  p10 {np.percentile(periods, 10):.1f}  p90 {np.percentile(periods, 90):.1f}")
tab = []
for (k, n), v in rows.items():
    if len(v) < 0.5 * len(periods):
        continue
    v = np.asarray([(a, b) for a, b, _ in v])
    tab.append((np.median(v[:, 0]), np.median(v[:, 1]), rows[(k, n)][0][2], k))
tab.sort()
ends = {}
print(f"{'start':>8} {'dur':>7} {'end':>8} {'gap':>6}  queue  kernel")
for s, d, q, k in tab:
    gap = s - ends.get(q, 0.0)
    print(f"{s:8.1f} {d:7.1f} {s + d:8.1f} {gap:6.1f}  {q:>5}  {k}")
    ends[q] = s + d
