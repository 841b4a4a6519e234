"""Print one step's kernel timeline from a rocprofv3 --kernel-trace csv (start/end in us relative to the step)."""
import sys, glob, re
import pandas as pd
f = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
df = pd.read_csv(f).sort_values("Start_Timestamp")
names = df.Kernel_Name.tolist()
# last occurrence of the first-layer forward kernel marks a step start
idx = [i for i, n in enumerate(names) if "conv_fwd_img_kernel<2, 2, true" in n or "conv_fwd_img_kernel<2, 1, true" in n]
start = idx[-2]; end = idx[-1]
t0 = df.iloc[start].Start_Timestamp
for i in range(start, end):
    r = df.iloc[i]
    n = re.sub(r"\(.*", "", r.Kernel_Name).replace("isdqn::", "").replace("void ", "")[:60]
    print(f"{(r.Start_Timestamp - t0)/1e3:8.1f} {(r.End_Timestamp - t0)/1e3:8.1f} {(r.End_Timestamp - r.Start_Timestamp)/1e3:7.1f}  q{r.Queue_Id} {n}")
