"""Bitwise run-to-run determinism of the forward activations and of a whole learn step at B=256 (debug aid)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "is-dqn_amd"))
import numpy as np, torch
from tests.gpu_helpers import make_frame_batch, device_batch
from slimdqn._engine import QNetEngine
B, K, A = 256, 9, 9
frames, ids, action, reward, terminal, _ = make_frame_batch(B, A, seed=5)
def mk():
    eng = QNetEngine((84, 84, 4), A, 1 + K, (32, 64, 64, 512), 'cnn', True, B, gamma_n=0.99, learning_rate=6.25e-5, adam_eps=1.5e-4)
    eng.init_params(1); return eng
eng = mk()
fr = torch.from_numpy(frames).cuda()
both = torch.from_numpy(np.concatenate([ids[:, :4], ids[:, 4:]], 0).copy()).cuda()
N = int(os.environ.get("REPS", "6"))
for name in ("act/Conv_0", "act/Conv_1", "act/Conv_2", "act/Dense_0"):
    ref = None; bad = 0
    for r in range(N):
        eng.forward(frames=fr, frame_stride=frames.shape[1], frame_ids=both, n_rows=2 * B); torch.cuda.synchronize()
        a = eng.region(name).clone()
        if ref is None: ref = a
        else: bad += int((a != ref).sum().item())
    print(f"{name:12s} differing elements over {N-1} repeats: {bad}")
# whole learn steps from identical state
outs = []
for r in range(4):
    e = mk(); b = device_batch(e, frames, ids, action, reward, terminal)
    for _ in range(3): e.learn_on_batch(b)
    torch.cuda.synchronize(); outs.append(e.params.clone())
print("params after 3 learn steps, differing elements vs run 0:", [int((o != outs[0]).sum().item()) for o in outs[1:]])
import itertools
print("pairwise differing params:", {(i, j): int((outs[i] != outs[j]).sum().item()) for i, j in itertools.combinations(range(len(outs)), 2)})
# which tensors
e = mk()
for name, off, size in [(n, o, s_) for n, o, s_ in e.param_layout()] if hasattr(e, "param_layout") else []:
    d = int((outs[0][off:off + size] != outs[1][off:off + size]).sum().item())
    if d: print("   ", name, d, "of", size)
