#!/bin/bash
# What one 5-step replay of the driver's form (bench.py --steps 20 --warmup 5) is made of besides its steps: kernel trace of the bench,
# gaps between the last kernel of a replay and the first convolution of the next.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/dft; rm -rf $out; mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $out/kt -- python3 bench.py --steps 400 --warmup 100 --graph 5 --no-cpu-baseline --replay-stats 0 > $out/kt.log 2>&1 || { tail -5 $out/kt.log; }
python3 - <<'PY'
import glob, re
import pandas as pd
f = glob.glob("gpurun_out/dft/kt/**/*_kernel_trace.csv", recursive=True)[0]
df = pd.read_csv(f).sort_values("Start_Timestamp").reset_index(drop=True)
df["k"] = df.Kernel_Name.map(lambda n: re.sub(r"\(.*", "", re.sub(r"^void ", "", n).replace("isdqn::", ""))[:48])
df = df.iloc[len(df) // 2:].reset_index(drop=True)
t0 = df.Start_Timestamp.iloc[0]
rows = []
for i, r in df.head(110).iterrows():
    rows.append(f"{(r.Start_Timestamp - t0) / 1e3:9.1f} {(r.End_Timestamp - t0) / 1e3:9.1f}  q{r.Queue_Id:<3} {r.k}")
open("gpurun_out/driver_form_trace.txt", "w").write("\n".join(rows) + "\n")
print("\n".join(rows[:100]))
PY
