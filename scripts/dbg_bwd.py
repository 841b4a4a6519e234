"""Run-to-run determinism of the backward intermediates of ONE learn step from identical state (debug aid)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "is-dqn_amd"))
import numpy as np, torch
from tests.gpu_helpers import make_frame_batch, device_batch
from slimdqn._engine import QNetEngine
B, K, A = 256, 9, 9
frames, ids, action, reward, terminal, _ = make_frame_batch(B, A, seed=5)
names = ["act/Conv_0", "act/Conv_2", "act/Dense_0", "dout", "dz/Dense_0", "dz/Conv_2", "dz/Conv_1", "dz/Conv_0", "gw/Conv_0", "gw/Conv_1", "gw/Conv_2", "gw/Dense_1", "red/Conv_0", "red/Conv_1", "red/Conv_2", "part/Dense_0"]
ref = None
tot = {n: 0 for n in names}; ptot = 0
R = int(os.environ.get("REPS", "8"))
for r in range(R):
    eng = QNetEngine((84, 84, 4), A, 1 + K, (32, 64, 64, 512), 'cnn', True, B, gamma_n=0.99, learning_rate=6.25e-5, adam_eps=1.5e-4)
    eng.init_params(1)
    b = device_batch(eng, frames, ids, action, reward, terminal)
    eng.learn_on_batch(b); torch.cuda.synchronize()
    cur = {}
    for n in names:
        try: cur[n] = eng.region(n).clone()
        except Exception as e: cur[n] = None
    cur["params"] = eng.params.clone()
    if ref is None: ref = cur
    else:
        for n in names:
            if cur[n] is not None: tot[n] += int((cur[n] != ref[n]).sum().item())
        ptot += int((cur["params"] != ref["params"]).sum().item())
for n in names: print(f"{n:14s} differing over {R-1} repeats: {tot[n]}")
print("params differing:", ptot)
