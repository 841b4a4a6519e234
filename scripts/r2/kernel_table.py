"""Per-kernel roofline table of one bench.py workload from a rocprofv3 kernel trace and two PMC passes (scripts/r2/profile.sh).

For every kernel of the step: launches per step, mean duration, counter traffic per step (2 x FETCH_SIZE + WRITE_SIZE,
KB units, gfx950 half-count correction on reads: MI355X_MICROARCH.md HBM section), ALGORITHMIC bytes and FLOPs of its
job (fp32 storage of what it must read and write once; MFMA passes of the split-bf16 scheme), and the fractions of
the 8 TB/s and 2.5 PF/s peaks they correspond to at the measured duration.  Non-step kernels (torch fills, copies) are
listed separately and excluded from the sums.

usage: kernel_table.py <profile dir> <workload> <timed steps of the PMC runs> <warm-up steps of the PMC runs>
"""
import glob
import json
import os
import re
import sys

import pandas as pd

d, wl, pmc_steps, pmc_warm = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
W = {"c2": (256, 9, 9), "c3": (256, 9, 9), "c5": (1024, 32, 4)}[wl]
B, K, A = W
NHA = (1 + K) * A
HBM, MFMA = 8.0e12, 2.5e15


def short(n):
    n = re.sub(r"^void ", "", n).replace("isdqn::", "")
    return re.sub(r"\(.*", "", n)[:110]


def is_step_kernel(k):
    return not (k.startswith("at::") or "rocclr" in k or k.startswith("void at::") or k == "")


# ---- algorithmic model: (bytes, flops, passes) per LAUNCH, by kernel-name pattern -----------------------------------
F4 = 4
a0, a1, a2 = 441 * 32, 121 * 64, 121 * 64          # activation elements per image
w0, w1, w2, wd, wh = 32 * 256, 64 * 512, 64 * 576, 7744 * 512, 512 * NHA
model = [
    (r"conv_fwd_img_kernel<2, 2, true|conv_fwd_u8_pair_kernel<2", "conv0 fwd + LN + ReLU", 2 * B * 28224 + 2 * B * a0 * F4 + B * a0 * F4 + w0 * F4, 2 * 2 * B * 441 * 32 * 256, 2),
    (r"conv_fwd_img_kernel<4, 3, false, 2|conv_fwd_s8_pair_kernel", "conv1 fwd + LN + ReLU", 2 * B * a0 * F4 + 2 * B * a1 * F4 + B * a1 * F4 + w1 * F4, 2 * 2 * B * 121 * 64 * 512, 3),
    (r"conv_fwd_img_kernel<4, 3, false, 1", "conv2 fwd + LN + ReLU", 2 * B * a1 * F4 + 2 * B * a2 * F4 + B * a2 * F4 + w2 * F4, 2 * 2 * B * 121 * 64 * 576, 3),
    (r"PlainGemm<128, (128|64), 2, 2, false, false, 3, true, false, false", "dense0 fwd (split-K slabs)", 2 * B * a2 * F4 + wd * F4 + 2 * B * 512 * F4, 2 * 2 * B * 7744 * 512, 3),
    (r"head_chain_kernel", "hidden LN + head GEMM + TD + head dgrad + LN bwd", 2 * B * 512 * F4 * 2 + wh * F4 + B * 512 * F4, 2 * 2 * B * 512 * NHA + 2 * B * K * 512, 3),
    (r"DenseDgradLN", "dense0 dgrad + LN/ReLU bwd of conv2", B * 512 * F4 + wd * F4 + B * a2 * F4 * 2, 2 * B * 7744 * 512, 3),
    (r"conv_dgrad_img_kernel<4, 3", "conv2 dgrad + LN/ReLU bwd of conv1", B * a2 * F4 + w2 * F4 + B * a1 * F4 * 2, 2 * B * 121 * 64 * 576, 3),
    (r"conv_dgrad_img_kernel<2, 3", "conv1 dgrad + LN/ReLU bwd of conv0", B * a1 * F4 + w1 * F4 + B * a0 * F4 * 2, 2 * B * 121 * 64 * 512, 3),
    (r"PlainGemm<64, 64, 2, 2, true, true, 3, true, false, true", "dense0 wgrad + fused Adam", B * 512 * F4 + B * a2 * F4 + 6 * wd * F4, 2 * B * 7744 * 512, 3),
    (r"conv_wgrad_img_kernel<4, 9, 3", "conv2 wgrad", B * a2 * F4 + B * a1 * F4 + w2 * F4, 2 * B * 121 * 64 * 576, 3),
    (r"conv_wgrad_img_kernel<4, 4, 3, false, 8", "conv1 wgrad", B * a1 * F4 + B * a0 * F4 + w1 * F4, 2 * B * 121 * 64 * 512, 3),
    (r"conv_wgrad_img_kernel<2, 4, 2, true", "conv0 wgrad", B * a0 * F4 + B * 28224 + w0 * F4, 2 * B * 441 * 32 * 256, 2),
    (r"PlainGemm<128, 128, 2, 2, true, true, 3, true, false, false", "head wgrad", B * NHA * F4 + B * 512 * F4 + wh * F4, 2 * B * 512 * NHA, 3),
    (r"adam_kernel", "Adam (conv / small tensors; slab reduction)", 7 * (w0 + w1 + w2 + wh) * F4, 0, 0),
    (r"reduce_rows_kernel", "LN-bwd partial-row reduction", 0, 0, 0),
    (r"loss_finalize_kernel", "loss / head-bias reduction", 0, 0, 0),
    (r"gather_rows_kernel", "replay row gather", B * (8 * 4 + 4 + 4 + 1 + 4), 0, 0),
    (r"tree_query_kernel", "sum-tree query (f64)", B * 21 * 16, 0, 0),
    (r"tree_set_kernel", "sum-tree set (f64)", B * 21 * 24, 0, 0),
]


def model_of(k):
    for pat, what, by, fl, ps in model:
        if re.search(pat, k):
            return what, by, fl, ps
    return "", 0, 0, 0


# ---- kernel trace -----------------------------------------------------------------------------------------------------
kt = glob.glob(os.path.join(d, "kt", "*", "*_kernel_trace.csv"))
df = pd.read_csv(kt[0]).sort_values("Start_Timestamp").reset_index(drop=True)
df["k"] = df.Kernel_Name.map(short)
df["dur"] = (df.End_Timestamp - df.Start_Timestamp) / 1e3
df = df.iloc[len(df) // 3:]  # steady state
anchor = df[df.k.str.contains("conv_fwd_img_kernel<2, 2, true|conv_fwd_u8_pair_kernel<2")]
n_steps = len(anchor)
per = (anchor.Start_Timestamp.values[1:] - anchor.Start_Timestamp.values[:-1]) / 1e3
per = per[per < 2 * pd.Series(per).median()]
g = df.groupby("k").dur.agg(["mean", "count"])
g["per_step"] = g["count"] / max(n_steps, 1)

# ---- PMC ---------------------------------------------------------------------------------------------------------------
def pmc(counter):
    f = glob.glob(os.path.join(d, "pmc_" + counter, "*", "*_counter_collection.csv"))
    if not f:
        return None
    t = pd.read_csv(f[0])
    t = t[t.Counter_Name == counter]
    t["k"] = t.Kernel_Name.map(short)
    return t.groupby("k").Counter_Value.sum() / (pmc_steps + pmc_warm)  # every launch of the run is counted: warm-up included


fe, wr = pmc("FETCH_SIZE"), pmc("WRITE_SIZE")
rows, tot = [], dict(us=0.0, mb=0.0, alg=0.0, fl=0.0)
other = []
for k, r in g.sort_values("mean", ascending=False).iterrows():
    if r["count"] < 0.1 * n_steps:
        other.append((k, r["mean"], int(r["count"])))
        continue
    if not is_step_kernel(k):
        other.append((k, r["mean"], int(r["count"])))
        continue
    what, by, fl, ps = model_of(k)
    n = r.per_step
    mb = None
    if fe is not None and wr is not None and k in fe.index:
        mb = (2 * fe.get(k, 0.0) + wr.get(k, 0.0)) * 1024 / 1e6
    us = r["mean"]
    rows.append((k, what, n, us, mb, by * n / 1e6, fl * n / 1e9, ps))
    tot["us"] += us * n
    tot["mb"] += (mb or 0.0)
    tot["alg"] += by * n / 1e6
    tot["fl"] += fl * n / 1e9

print(f"# {wl}: per-kernel roofline table (B={B}, K={K}, A={A}; split-bf16), steady state of bench.py under rocprofv3 --kernel-trace\n")
try:
    print("bench line under trace: `" + open(os.path.join(d, "bench_under_trace.json")).read().strip()[:400] + " ...`\n")
except OSError:
    pass
print("| kernel | job | launches/step | mean us | counter MB/step | algorithmic MB/step | counter/alg | HBM frac (alg bytes / us / 8 TB/s) | GFLOP/step | MFMA frac (flops x passes / us / 2.5 PF) |")
print("|---|---|---|---|---|---|---|---|---|---|")
for k, what, n, us, mb, alg, fl, ps in rows:
    hb = alg * 1e6 / (us * n * 1e-6) / HBM if alg and us else 0.0
    mf = fl * 1e9 * ps / (us * n * 1e-6) / MFMA if fl and us else 0.0
    ratio = f"{mb / alg:.2f}" if (mb and alg) else ""
    print(f"| `{k}` | {what} | {n:.2f} | {us:.1f} | {'' if mb is None else f'{mb:.1f}'} | {alg:.1f} | {ratio} | {hb:.3f} | {fl:.2f} | {mf:.3f} |")
print(f"| **sum over the step** | | | **{tot['us']:.1f}** | **{tot['mb']:.1f}** | **{tot['alg']:.1f}** | {tot['mb'] / max(tot['alg'], 1e-9):.2f} | | **{tot['fl']:.1f}** | |")
print(f"\nstep period (start to start of the first kernel, two streams overlap): **{per.mean():.1f} us** over {n_steps} steps; kernel time summed: {tot['us']:.1f} us")
if other:
    print("\nnot part of the step (setup, copies; excluded above): " + "; ".join(f"`{k}` {c} x {m:.1f} us" for k, m, c in other[:8]))
json.dump({"workload": wl, "precision": "bf16x3", "hbm_bytes_per_step_corrected": tot["mb"] * 1e6, "build": os.environ.get("ISDQN_BUILD_TAG"),
           "command": f"rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE -- python3 bench.py --workload {wl} --no-cpu-baseline --steps {pmc_steps} --warmup {pmc_warm} (step kernels only; 2*FETCH_SIZE + WRITE_SIZE, KB units)",
           "step_period_us": float(per.mean()), "kernel_time_sum_us": tot["us"], "algorithmic_MB_sum": tot["alg"]},
          open(os.path.join(d, f"{wl}_hbm_traffic.json"), "w"), indent=1)
