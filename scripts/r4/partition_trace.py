"""Raw view of a kernel trace with several replicas: per queue the kernel count, busy time, and a window of consecutive kernels with
start / end / queue so that overlap (or its absence) between the replicas' queues is visible.  usage: partition_trace.py <kernel_trace.csv>"""
import re
import sys

import pandas as pd

df = pd.read_csv(sys.argv[1]).sort_values("Start_Timestamp").reset_index(drop=True)
df["k"] = df.Kernel_Name.map(lambda n: re.sub(r"\(.*", "", re.sub(r"^void ", "", n).replace("isdqn::", ""))[:60])
df = df.iloc[len(df) // 2:].reset_index(drop=True)
t0 = df.Start_Timestamp.iloc[0]
span = (df.End_Timestamp.max() - t0) / 1e3
print(f"{len(df)} kernels over {span:.0f} us")
for q, g in df.groupby("Queue_Id"):
    busy = ((g.End_Timestamp - g.Start_Timestamp).sum()) / 1e3
    print(f"queue {q}: {len(g)} kernels, busy {busy:.0f} us ({100 * busy / span:.0f} % of the span), mean duration {busy / len(g):.1f} us")
print("\nfirst 70 kernels of the window: start, end (us), queue, kernel")
for _, r in df.head(70).iterrows():
    print(f"{(r.Start_Timestamp - t0) / 1e3:9.1f} {(r.End_Timestamp - t0) / 1e3:9.1f}  q{r.Queue_Id:<3} {r.k}")
