import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "is-dqn_amd"))
import numpy as np, torch
from tests.gpu_helpers import make_frame_batch, device_batch
from slimdqn._engine import QNetEngine
B, K, A = 256, 9, 9
frames, ids, action, reward, terminal, _ = make_frame_batch(B, A, seed=5)
def run(nsteps, sync):
    eng = QNetEngine((84, 84, 4), A, 1 + K, (32, 64, 64, 512), 'cnn', True, B, gamma_n=0.99, learning_rate=6.25e-5, adam_eps=1.5e-4)
    eng.init_params(1)
    b = device_batch(eng, frames, ids, action, reward, terminal)
    for _ in range(nsteps):
        eng.learn_on_batch(b)
        if sync: torch.cuda.synchronize()
    torch.cuda.synchronize()
    return eng.params.clone(), eng.adam_m.clone()
for nsteps in (1, 2, 3, 6):
    for sync in (True, False):
        outs = [run(nsteps, sync) for _ in range(5)]
        print(f"steps={nsteps} sync_between={sync}: params differing vs run0:", [int((o[0] != outs[0][0]).sum().item()) for o in outs[1:]])
