"""In-kernel phase stamps of the image-resident forward conv kernel of one layer (debug hook), inside a full
learn step.  Prints per-phase medians in microseconds (s_memtime ticks are shader cycles; the clock is derived
from kernel-wide span vs s_memrealtime at 100 MHz)."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "is-dqn_amd"))
import numpy as np, torch
from slimdqn._engine import QNetEngine
from slimdqn import _hip

layer = sys.argv[1] if len(sys.argv) > 1 else "Conv_1"
B, K, A = int(os.environ.get("B", "256")), 9, 9
eng = QNetEngine((84, 84, 4), A, 1 + K, (32, 64, 64, 512), "cnn", True, B, gamma_n=0.99, learning_rate=6.25e-5, adam_eps=1.5e-4)
eng.init_params(0)
nf = 1_000_000
g = torch.Generator(device="cuda").manual_seed(0)
frames = torch.randint(0, 256, (nf, 84 * 84), dtype=torch.uint8, device="cuda", generator=g)
base = torch.randint(0, nf - 8, (B, 1), device="cuda", generator=g)
ids = (base + torch.arange(4, device="cuda")[None, :]).int()
ids = torch.cat([ids, ids + 1], 1).contiguous()
batch = eng.make_batch(frames=frames, frame_stride=84 * 84, frame_ids=ids, action=torch.randint(0, A, (B,), device="cuda", generator=g).int(),
                       reward=torch.randn(B, device="cuda", generator=g), terminal=torch.zeros(B, dtype=torch.uint8, device="cuda"))
for _ in range(5):
    eng.learn_on_batch(batch)
nwg = 4096
st = torch.zeros(nwg * 8, dtype=torch.int64, device="cuda")
lib = _hip.lib()
lib.isdqn_debug_set_stamps.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
lib.isdqn_debug_set_stamps(ctypes.c_void_p(st.data_ptr()), layer.encode())
eng.learn_on_batch(batch)
torch.cuda.synchronize()
lib.isdqn_debug_set_stamps(None, b"")
s = st.cpu().numpy().reshape(nwg, 8)
s = s[s[:, 7] != 0]  # column 7 (s_memrealtime) is written by the workgroup-level stamp 0 only
print(layer, "workgroups", len(s))
t0 = s[:, 0].min()
span_cyc = s[:, 5].max() - t0
span_rt = (s[:, 7].max() - s[:, 7].min())  # start-to-last-start only; use as lower bound
order = np.argsort(s[:, 0])
names = ["fill(load+convert+write)", "weights0+sync", "K loop", "epilogue issue", "store drain"]
if layer.startswith("wgrad:"):
    names = ["image 0 staged (fetch + commit)", "K steps of image 0", "image 1 staged (commit)", "K steps of image 1", "slab stores drained"]
if layer == "head_chain":
    names = ["operand requests + hidden rows -> LDS", "head GEMM", "q reduce + targets/TD", "data-gradient AXPY", "LayerNorm backward"]
print("(layer may be e.g. Conv_1 for the forward kernel or dgrad:Conv_1 for the data-gradient kernel of that layer)")
d = np.diff(s[:, :6], axis=1)
ghz = float(os.environ.get("GHZ", "2.0"))
# s_memtime counters are per XCD (not comparable across workgroups); the 100 MHz s_memrealtime of stamp 0 is global
rt = (s[:, 7] - s[:, 7].min()) / 100.0
print("workgroup start, us after the first (realtime): p10 %.2f p25 %.2f p50 %.2f p75 %.2f p90 %.2f max %.2f" % tuple(np.percentile(rt, [10, 25, 50, 75, 90, 100])))
print("start histogram (2 us bins):", np.histogram(rt, bins=np.arange(0, rt.max() + 2, 2))[0].tolist())
for i, n in enumerate(names):
    print(f"{n:28s} median {np.median(d[:, i]):9.0f} cyc  p10 {np.percentile(d[:, i], 10):9.0f}  p90 {np.percentile(d[:, i], 90):9.0f}   ~{np.median(d[:, i]) / ghz / 1e3:6.2f} us @ {ghz} GHz")
# first-round workgroups run the code cold (instruction cache), later rounds find it warm
first = rt < 2.0
if first.sum() and (~first).sum():
    for i, n in enumerate(names):
        print(f"   {n:28s} first-round median {np.median(d[first, i]):9.0f}   later rounds {np.median(d[~first, i]):9.0f}")
    tot = s[:, 5] - s[:, 0]
    print(f"   per-WG total: first-round {np.median(tot[first]):.0f}  later rounds {np.median(tot[~first]):.0f}")
if layer == "head_chain" and s[:, 6].any():
    print(f"   of the first phase: slab sums + bias until {np.median(s[:, 6] - s[:, 0]):.0f} cyc after kernel start (thread 0)")
print("per-WG total median", np.median(s[:, 5] - s[:, 0]), "cyc; kernel span", span_cyc, "cyc")
# first-round vs second-round workgroups
# optional step-level stamps (diagnostic build only): second row block
s2 = st.cpu().numpy().reshape(nwg, 8)
g = len(s)
rows = s2[g:2 * g]
if rows[:, 0].any():
    dd = np.diff(rows[:, :6], axis=1)
    for i, n in enumerate(["fetch issue", "stash (vmcnt wait + cvt + ds_write)", "barrier (lgkmcnt(0) + s_barrier)", "read_frags issue", "24 MFMAs issue"]):
        print(f"  step: {n:40s} median {np.median(dd[:, i]):7.0f} cyc  p90 {np.percentile(dd[:, i], 90):7.0f}")
