"""Per-kernel mean of rocprofv3 --pmc counters (counter_collection.csv), one row per kernel name."""
import sys, glob, re
import pandas as pd
f = glob.glob(sys.argv[1] + "/*/*_counter_collection.csv")[0]
df = pd.read_csv(f)
df["k"] = df.Kernel_Name.map(lambda n: re.sub(r"\(.*", "", n).replace("isdqn::", "").replace("void ", "")[:58])
df["dur"] = (df.End_Timestamp - df.Start_Timestamp) / 1e3
t = df.pivot_table(index="k", columns="Counter_Name", values="Counter_Value", aggfunc="mean")
t["n"] = df.groupby("k").Dispatch_Id.nunique()
t["grid"] = df.groupby("k").Grid_Size.first() // df.groupby("k").Workgroup_Size.first()
pd.set_option("display.width", 250); pd.set_option("display.max_columns", 30); pd.set_option("display.float_format", lambda x: f"{x:,.0f}")
print(t.sort_values(t.columns[0], ascending=False).head(24).to_string())
