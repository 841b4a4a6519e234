"""Timeline of the last full step in a rocprofv3 kernel trace of bench.py (graph replay)."""
import sys, glob, re
import pandas as pd
f = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
df = pd.read_csv(f).sort_values("Start_Timestamp").reset_index(drop=True)
names = df.Kernel_Name.tolist()
idx = [i for i, n in enumerate(names) if "conv_fwd_img_kernel<2, 2, true" in n]
print("n steps seen", len(idx))
k = int(sys.argv[2]) if len(sys.argv) > 2 else -3
start, end = idx[k], idx[k + 1]
t0 = df.iloc[start].Start_Timestamp
for i in range(max(0, start - 6), end + 1):
    r = df.iloc[i]
    n = re.sub(r"\(.*", "", r.Kernel_Name).replace("isdqn::", "").replace("void ", "")[:64]
    print(f"{(r.Start_Timestamp - t0)/1e3:8.1f} {(r.End_Timestamp - t0)/1e3:8.1f} {(r.End_Timestamp - r.Start_Timestamp)/1e3:7.1f}  q{r.Queue_Id} {n}")
d = [(df.iloc[idx[j + 1]].Start_Timestamp - df.iloc[idx[j]].Start_Timestamp) / 1e3 for j in range(len(idx) - 1)]
print("step periods (us):", [round(x, 1) for x in d[-24:]])
