"""Localise gradient deviations: HIP intermediates (z, dz per layer) vs a float64 torch forward/backward."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "is-dqn_amd"))
import numpy as np, torch, torch.nn.functional as F
from tests.gpu_helpers import device_batch, make_frame_batch, make_pair
feats = tuple(int(x) for x in os.environ.get("FEATS", "32,64,64,512").split(","))
K, A, B = int(os.environ.get("K", "9")), int(os.environ.get("A", "9")), int(os.environ.get("B", "32"))
oracle, eng, params = make_pair(feats, K, A, B, layer_norm=True, seed=7)
frames, ids, action, reward, terminal, ref = make_frame_batch(B, A, seed=23, n_frames=B + 64)
batch = device_batch(eng, frames, ids, action, reward, terminal)
grad = torch.zeros_like(eng.params)
eng.learn_on_batch(batch, grad_out=grad); torch.cuda.synchronize()
P = {m: {k: torch.tensor(np.asarray(v), dtype=torch.float64) for k, v in d.items()} for m, d in params.items()}
def same(size, k, s):
    out = -(-size // s); tot = max((out - 1) * s + k - size, 0); return tot // 2, tot - tot // 2
x = torch.tensor(np.concatenate([ref.state, ref.next_state]), dtype=torch.float64) / 255.0
x = x.permute(0, 3, 1, 2).contiguous().requires_grad_(True)
zs = []
for i, (k, s) in enumerate(((8, 4), (4, 2), (3, 1))):
    lo, hi = same(x.shape[2], k, s)
    xp = F.pad(x, (lo, hi, lo, hi))
    w = P[f"Conv_{i}"]["kernel"].permute(3, 2, 0, 1)
    z = F.conv2d(xp, w, P[f"Conv_{i}"]["bias"], stride=s)
    z.retain_grad(); zs.append(z)
    zc = z.permute(0, 2, 3, 1)
    mean = zc.mean(-1, keepdim=True); var = ((zc * zc).mean(-1, keepdim=True) - mean * mean).clamp_min(0)
    y = (zc - mean) * torch.rsqrt(var + 1e-6) * P[f"LayerNorm_{i}"]["scale"] + P[f"LayerNorm_{i}"]["bias"]
    x = torch.relu(y).permute(0, 3, 1, 2)
h = x.permute(0, 2, 3, 1).reshape(2 * B, -1)
zd = h @ P["Dense_0"]["kernel"] + P["Dense_0"]["bias"]; zd.retain_grad(); zs.append(zd)
mean = zd.mean(-1, keepdim=True); var = ((zd * zd).mean(-1, keepdim=True) - mean * mean).clamp_min(0)
hd = torch.relu((zd - mean) * torch.rsqrt(var + 1e-6) * P["LayerNorm_3"]["scale"] + P["LayerNorm_3"]["bias"])
q = (hd @ P["Dense_1"]["kernel"] + P["Dense_1"]["bias"]).reshape(2 * B, 1 + K, A)
act = torch.tensor(action, dtype=torch.long)
qv = q[:B, 1:, :][torch.arange(B), :, act]
tg = torch.tensor(reward, dtype=torch.float64)[:, None] + (1 - torch.tensor(terminal, dtype=torch.float64))[:, None] * 0.99 * q[B:, :K].max(-1).values
loss = ((qv - tg.detach()) ** 2).mean(0).sum()
loss.backward()
names = ["Conv_0", "Conv_1", "Conv_2", "Dense_0"]
for i, n in enumerate(names):
    zr = zs[i].detach()[:B]; gr = zs[i].grad[:B]
    if i < 3:
        zr = zr.permute(0, 2, 3, 1); gr = gr.permute(0, 2, 3, 1)
    zr = zr.reshape(B, -1).numpy(); gr = gr.reshape(B, -1).numpy()
    zh = eng.region("z/" + n)[: zr.size].cpu().numpy().reshape(B, -1).astype(np.float64)
    gh = eng.region("dz/" + n)[: gr.size].cpu().numpy().reshape(B, -1).astype(np.float64)
    ez = np.abs(zh - zr); eg = np.abs(gh - gr)
    per_img = eg.max(1) / np.abs(gr).max()
    worst = np.argsort(-per_img)[:4]
    print(f"{n}: z max err {ez.max():.1e} (|z| {np.abs(zr).max():.1f})  dz max err {eg.max():.2e} rel {eg.max()/np.abs(gr).max():.1e}  worst images {[(int(w), float('%.1e' % per_img[w])) for w in worst]}")
    w0 = worst[0]
    C = feats[i] if i < 3 else feats[3]
    e = eg[w0].reshape(-1, C)
    pix = np.argsort(-e.max(1))[:6]
    print("    worst image", int(w0), "pixels", [(int(p), float('%.1e' % e[p].max()), int((e[p] > 0.1 * e.max()).sum())) for p in pix], "n elements > 10% of max:", int((e > 0.1 * e.max()).sum()))
