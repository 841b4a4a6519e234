"""gradient of the residual stream inside the impala torso: HIP workspace buffers against torch autograd (diagnostic)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "is-dqn_amd"))
import numpy as np, torch
from oracle import network as net
from tests.gpu_helpers import device_batch, make_frame_batch, make_pair
feats, K, A, B, ln, obs = (8, 16, 16, 24), 2, 5, 4, True, (84, 84, 4)
oracle, eng, params = make_pair(feats, K, A, B, arch="impala", obs=obs, layer_norm=ln, seed=3, lr=1e-3)
frames, ids, action, reward, terminal, ref = make_frame_batch(B, A, seed=7)
batch = device_batch(eng, frames, ids, action, reward, terminal)
g = torch.zeros_like(eng.params)
eng.grad_on_batch(batch, g)
torch.cuda.synchronize()
# oracle with captured stream tensors
P = {m: {n: t.detach().clone().requires_grad_(True) for n, t in l.items()} for m, l in oracle.params.items()}
cap = {}
state, action_t, reward_t, next_state, terminal_t = oracle._batch_tensors(ref)
x = torch.cat((state, next_state))
q = net.forward(P, x, feats, "impala", ln, cap).reshape(-1, 1 + K, A)
for v in cap.values():
    if v.requires_grad: v.retain_grad()
qv = q[:B, 1:].gather(2, action_t.view(B, 1, 1).expand(B, K, 1)).squeeze(2)
tg = oracle.compute_target(reward_t[:, None], terminal_t[:, None], q[B:, :-1]).detach()
((qv - tg) ** 2).mean(0).sum().backward()
for s in range(3):
    for name in (f"imp/s{s}/r0", f"imp/s{s}/r1", f"imp/s{s}/r2"):
        key = f"Stack_{s}/r{name[-1]}" if name[-1] != "2" else f"Stack_{s}"
        t = cap[key]
        hip = eng.region(name).cpu().numpy()[: t.numel()].reshape(t.shape)
        print(name, "forward max err", float(np.abs(hip - t.detach().numpy()).max()))
    # dr buffer of stack s after the backward = gradient w.r.t. r0 (pooled) of stack s, first B images
    src = f"imp/s{s}/dr" if s == 2 else f"imp/s{s+1}/da"
    gr = cap[f"Stack_{s}/r0"].grad[:B].numpy()
    hip = eng.region(src).cpu().numpy()[: gr.size].reshape(gr.shape)
    err = np.abs(hip - gr)
    per_img = [float(np.linalg.norm(hip[b] - gr[b]) / np.linalg.norm(gr[b])) for b in range(B)]
    print(f"d r0 of stack {s}: rel err per image {per_img}")
    rows = err.reshape(B, -1, err.shape[-1]).max(-1)
    bad = np.argwhere(rows > 1e-3 * np.abs(gr).max())
    print("   rows with large error:", len(bad), bad[:12].tolist())
