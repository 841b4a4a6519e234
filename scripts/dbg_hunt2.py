import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "is-dqn_amd"))
import numpy as np, torch
from tests.gpu_helpers import make_frame_batch, device_batch
from slimdqn._engine import QNetEngine
B, K, A = int(os.environ.get("B", "256")), int(os.environ.get("K", "9")), int(os.environ.get("A", "9"))
frames, ids, action, reward, terminal, _ = make_frame_batch(B, A, seed=13)
def mk():
    e = QNetEngine((84, 84, 4), A, 1 + K, (32, 64, 64, 512), 'cnn', True, B, gamma_n=0.99, learning_rate=6.25e-5, adam_eps=1.5e-4)
    return e
mode = os.environ.get("MODE", "same")   # same: one engine reset each repeat; fresh: new engine each repeat; poison: fresh + workspace filled with NaN-free garbage
R = int(os.environ.get("REPS", "40")); STEPS = int(os.environ.get("STEPS", "4")); SYNC = os.environ.get("SYNC", "0") == "1"
eng = mk(); b = device_batch(eng, frames, ids, action, reward, terminal)
ref = None; events = 0
for r in range(R):
    if mode != "same":
        eng = mk(); b = device_batch(eng, frames, ids, action, reward, terminal)
        if mode == "poison": eng.workspace.uniform_(-3.0, 3.0)
    eng.init_params(3); eng.adam_m.zero_(); eng.adam_v.zero_(); eng.adam_count.zero_()
    for _ in range(STEPS):
        eng.learn_on_batch(b)
        if SYNC: torch.cuda.synchronize()
    torch.cuda.synchronize()
    cur = {"params": eng.params.clone(), "adam_m": eng.adam_m.clone(), "adam_v": eng.adam_v.clone()}
    if ref is None: ref = cur; continue
    d = [(n, int((cur[n] != ref[n]).sum().item())) for n in cur if not torch.equal(cur[n], ref[n])]
    if d:
        events += 1
        print(f"  repeat {r}: " + ", ".join(f"{n}:{c}" for n, c in d))
print(f"mode={mode} steps={STEPS} sync={SYNC}: {events} differing repeats out of {R-1}")
