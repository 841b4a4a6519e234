"""Phase stamps of conv_fwd_u8_pair_kernel (dev build): per workgroup start (s_memrealtime), tile-0 image written, K loop, epilogue,
tile-1 image written, K loop, epilogue.  usage: ISDQN_HIP_LIB=.../libisdqn_hip_dev.so python scripts/r3/stamps_pair.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "is-dqn_amd"))
import numpy as np, torch
from slimdqn._engine import QNetEngine
from slimdqn import _hip

B, K, A = 256, 9, 9
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
lib.isdqn_debug_set_stamps(ctypes.c_void_p(st.data_ptr()), b"Conv_0")
eng.learn_on_batch(batch)
torch.cuda.synchronize()
lib.isdqn_debug_set_stamps(None, b"")
s = st.cpu().numpy().reshape(nwg, 8)
s = s[s[:, 7] != 0]
print("workgroups", len(s))
rt = (s[:, 7] - s[:, 7].min()) / 100.0
print("workgroup start, us after the first: p10 %.2f p50 %.2f p90 %.2f max %.2f" % tuple(np.percentile(rt, [10, 50, 90, 100])))
ghz = 2.4
names = ["fill 0 (request + wait + write)", "K loop 0", "epilogue 0", "commit 1 (wait + write)", "K loop 1", "epilogue 1"]
d = np.diff(s[:, :7], axis=1)
for i, n in enumerate(names):
    print(f"{n:34s} median {np.median(d[:, i]):8.0f} cyc  p10 {np.percentile(d[:, i], 10):8.0f}  p90 {np.percentile(d[:, i], 90):8.0f}  ~{np.median(d[:, i]) / ghz / 1e3:5.2f} us")
print("per workgroup total median %.0f cyc ~%.2f us" % (np.median(s[:, 6] - s[:, 0]), np.median(s[:, 6] - s[:, 0]) / ghz / 1e3))
