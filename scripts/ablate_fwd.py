"""Times the forward pass (512 images) under ISDQN_ABLATE flags, per kernel via torch events around forward."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "is-dqn_amd"))
import torch
from slimdqn._engine import QNetEngine
B, K, A = 256, 9, 9
eng = QNetEngine((84, 84, 4), A, 1 + K, (32, 64, 64, 512), "cnn", True, B, precision=sys.argv[1] if len(sys.argv) > 1 else "bf16x3")
eng.init_params(0)
nf = 20000
g = torch.Generator(device="cuda").manual_seed(0)
frames = torch.randint(0, 256, (nf, 84 * 84), dtype=torch.uint8, device="cuda", generator=g)
ids = torch.randint(0, nf, (2 * B, 4), device="cuda", generator=g).int().contiguous()
for _ in range(5):
    eng.forward(frames=frames, frame_stride=84 * 84, frame_ids=ids, n_rows=2 * B)
torch.cuda.synchronize()
t = time.perf_counter()
n = 50
for _ in range(n):
    eng.forward(frames=frames, frame_stride=84 * 84, frame_ids=ids, n_rows=2 * B)
torch.cuda.synchronize()
print(f"ablate={os.environ.get('ISDQN_ABLATE','0')} forward_us={(time.perf_counter()-t)/n*1e6:.1f}")
