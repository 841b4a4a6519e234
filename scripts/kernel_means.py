"""Mean duration per kernel (us) and mean step period from a rocprofv3 --kernel-trace csv of bench.py.
Usage: kernel_means.py <dirA> [<dirB>]  -> side-by-side table."""
import sys, glob, re
import pandas as pd

def load(d):
    f = glob.glob(d + "/*/*_kernel_trace.csv")[0]
    df = pd.read_csv(f).sort_values("Start_Timestamp").reset_index(drop=True)
    df["k"] = df.Kernel_Name.map(lambda n: re.sub(r"\(.*", "", n).replace("isdqn::", "").replace("void ", "")[:62])
    df["dur"] = (df.End_Timestamp - df.Start_Timestamp) / 1e3
    starts = df[df.k.str.contains("conv_fwd_img_kernel<2, 2, true|conv_fwd_u8_pair_kernel<2")].Start_Timestamp.values
    per = (starts[1:] - starts[:-1]) / 1e3
    per = per[per < 2 * pd.Series(per).median()]
    # steady state only: drop the first third
    n0 = len(df) // 3
    g = df.iloc[n0:].groupby("k").dur.agg(["mean", "count"])
    return g, per[len(per) // 3:].mean()

a, pa = load(sys.argv[1])
if len(sys.argv) > 2:
    b, pb = load(sys.argv[2])
    t = a.join(b, lsuffix="_A", rsuffix="_B", how="outer")
    t = t[t["count_A"].fillna(0) + t["count_B"].fillna(0) > 50].sort_values("mean_A", ascending=False)
    t["B/A"] = t.mean_B / t.mean_A
    pd.set_option("display.width", 200)
    print(t[["mean_A", "mean_B", "B/A"]].round(2).to_string())
    print(f"sum A {t.mean_A.sum():.1f}  sum B {t.mean_B.sum():.1f}   step period A {pa:.1f} us  B {pb:.1f} us")
else:
    a = a[a["count"] > 50].sort_values("mean", ascending=False)
    print(a.round(2).to_string())
    print(f"sum {a['mean'].sum():.1f}  step period {pa:.1f} us")
