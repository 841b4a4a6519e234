"""per-leaf gradient error of the impala torso against the oracle (diagnostic)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "is-dqn_amd"))
import numpy as np, torch
from tests.gpu_helpers import device_batch, make_frame_batch, make_pair
feats, K, A, B, ln, obs = (8, 16, 16, 24), 2, 5, int(os.environ.get("B", "4")), os.environ.get("LN", "1") == "1", (84, 84, 4)
oracle, eng, params = make_pair(feats, K, A, B, arch="impala", obs=obs, layer_norm=ln, seed=3, lr=1e-3)
frames, ids, action, reward, terminal, ref = make_frame_batch(B, A, seed=7)
batch = device_batch(eng, frames, ids, action, reward, terminal)
g = torch.zeros_like(eng.params)
eng.grad_on_batch(batch, g)
o_grads, _ = oracle.grads(oracle.params, ref)
hip_g = eng.internal_to_flax_grads(g)
for mod in o_grads:
    for leaf in o_grads[mod]:
        a, b = np.asarray(hip_g[mod][leaf], np.float64), o_grads[mod][leaf].numpy().astype(np.float64)
        print(f"{mod:24s} {leaf:8s} rel err {np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30):.2e}   |b| {np.linalg.norm(b):.3e}")
