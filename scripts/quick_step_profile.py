"""Quick timing of learn_on_batch at the headline shape (B=256, K=9, A=9, cnn 32/64/64/512)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "is-dqn_amd"))
import numpy as np, torch
from slimdqn._engine import QNetEngine

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
B, K, A = 256, 9, 9
eng = QNetEngine((84, 84, 4), A, 1 + K, (32, 64, 64, 512), "cnn", True, B, gamma_n=0.99, learning_rate=6.25e-5, adam_eps=1.5e-4, precision=prec)
eng.init_params(0)
nf = 100_000
g = torch.Generator(device="cuda").manual_seed(0)
frames = torch.randint(0, 256, (nf, 84 * 84), dtype=torch.uint8, device="cuda", generator=g)
base = torch.randint(0, nf - 8, (B, 1), device="cuda", generator=g)
ids = (base + torch.arange(4, device="cuda")[None, :]).int()
ids = torch.cat([ids, ids + 1], 1).contiguous()
action = torch.randint(0, A, (B,), device="cuda", generator=g).int()
reward = torch.randn(B, device="cuda", generator=g)
term = (torch.rand(B, device="cuda", generator=g) < 0.005).to(torch.uint8)
batch = eng.make_batch(frames=frames, frame_stride=84 * 84, frame_ids=ids, action=action, reward=reward, terminal=term)
for _ in range(5):
    eng.learn_on_batch(batch)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(steps):
    eng.learn_on_batch(batch)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / steps
print(f"precision={prec} ms_per_step={dt*1e3:.3f} steps_per_s={1/dt:.1f} losses={eng.losses.cpu().numpy()[:3]}")
