"""Explain the unstable dgamma partial sums of the 64-row DenseDgradLN: per failing element, which rows' terms are
missing / doubled relative to an fp64 host emulation."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "is-dqn_amd"))
import numpy as np, torch
from tests.gpu_helpers import make_frame_batch, device_batch
from slimdqn._engine import QNetEngine
B, K, A = 256, 9, 9
frames, ids, action, reward, terminal, _ = make_frame_batch(B, A, seed=5)
eng = QNetEngine((84, 84, 4), A, 1 + K, (32, 64, 64, 512), 'cnn', True, B, gamma_n=0.99, learning_rate=6.25e-5, adam_eps=1.5e-4)
b = device_batch(eng, frames, ids, action, reward, terminal)
def xcd_remap(bid, n):
    xcd, q, r = bid & 7, n >> 3, n & 7
    base = xcd * (q + 1) if xcd < r else r * (q + 1) + (xcd - r) * q
    return base + (bid >> 3)
runs = []
for r in range(8):
    eng.init_params(1); eng.adam_m.zero_(); eng.adam_v.zero_(); eng.adam_count.zero_()
    if r == 0:
        flax = eng.export_flax()
    eng.learn_on_batch(b); torch.cuda.synchronize()
    runs.append({n: eng.region(n).clone().cpu().numpy() for n in ["part/Conv_2", "dz/Conv_2", "z/Conv_2", "dz/Dense_0"]})
W = torch.tensor(np.asarray(flax["Dense_0"]["kernel"]), dtype=torch.float64)      # (7744, 512)
gam = np.asarray(flax["LayerNorm_2"]["scale"], np.float64); bet = np.asarray(flax["LayerNorm_2"]["bias"], np.float64)
dzd = torch.tensor(runs[0]["dz/Dense_0"][: B * 512].reshape(B, 512), dtype=torch.float64)
da = (dzd @ W.T).numpy().reshape(B, 121, 64)
z = runs[0]["z/Conv_2"][: B * 7744].reshape(B, 121, 64).astype(np.float64)
mean = z.mean(-1, keepdims=True); var = np.maximum((z * z).mean(-1, keepdims=True) - mean * mean, 0)
xh = (z - mean) / np.sqrt(var + 1e-6)
dy = np.where(xh * gam + bet > 0, da, 0.0)
term = dy * xh                                   # [b][pix][ch]: dgamma contributions
rows = 484
parts = [r_["part/Conv_2"][: rows * 192].reshape(rows, 3, 64) for r_ in runs]
stack = np.stack(parts)                          # [run][wg][3][64]
unstable = np.argwhere((stack != stack[0]).any(0))
print(len(unstable), "unstable elements; which:", np.bincount(unstable[:, 1], minlength=3))
shown = 0
for (wg, which, c) in unstable:
    bid = xcd_remap(int(wg), rows); m0 = (bid % 4) * 64; pix = bid // 4
    t = term[m0:m0 + 64, pix, c]                 # 64 row terms; wave w rows 16w..16w+15; lane group g rows 4g..4g+3 within the wave
    exp = t.sum()
    vals = stack[:, wg, which, c].astype(np.float64)
    msg = []
    for v in sorted(set(vals.tolist())):
        d = v - exp
        # does the deviation equal minus one row term / one 4-row group / one wave?
        best = None
        for name, cands in (("row", t), ("grp", t.reshape(16, 4).sum(1)), ("wave", t.reshape(4, 16).sum(1))):
            for i, x in enumerate(cands):
                for sgn in (-1, +1):
                    e = abs(d - sgn * x)
                    if best is None or e < best[0]:
                        best = (e, f"{'+' if sgn > 0 else '-'}{name}{i}")
        msg.append(f"{v:+.6e} (dev {d:+.2e}, nearest {best[1]} resid {best[0]:.1e}, n={int((vals == v).sum())})")
    print(f"wg {wg} bid {bid} m0 {m0} pix {pix} ch {c}: exp {exp:+.6e}; " + "; ".join(msg))
    shown += 1
    if shown >= 40: break
