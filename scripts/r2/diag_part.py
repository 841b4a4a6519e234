"""Which elements of part/Conv_2 (DenseDgradLN partial sums) differ between identical steps, and by how much."""
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
runs = []
for r in range(6):
    eng.init_params(1); eng.adam_m.zero_(); eng.adam_v.zero_(); eng.adam_count.zero_()
    eng.learn_on_batch(b); torch.cuda.synchronize()
    runs.append({n: eng.region(n).clone().cpu().numpy() for n in ["part/Conv_2", "dz/Conv_2", "z/Conv_2", "dz/Dense_0"]})
rows = int(os.environ.get("ROWS", "484"))
p0 = runs[0]["part/Conv_2"][: rows * 192].reshape(rows, 3, 64)
for r in range(1, 6):
    p = runs[r]["part/Conv_2"][: rows * 192].reshape(rows, 3, 64)
    d = np.argwhere(p != p0)
    print(f"run {r}: {len(d)} differing; dz equal {np.array_equal(runs[r]['dz/Conv_2'], runs[0]['dz/Conv_2'])}")
    for (wg, which, c) in d[:25]:
        a, bb = p0[wg, which, c], p[wg, which, c]
        print(f"   wg {wg} (m-tile {wg % 4 if rows == 484 else wg % 2}, n-tile {wg // 4 if rows == 484 else wg // 2}) which {which} ch {c}: {a!r} vs {bb!r}  rel {abs(a-bb)/max(abs(a),1e-30):.2e}")
    if len(d):
        print("   by which:", np.bincount(d[:, 1], minlength=3), " channels:", sorted(set(d[:, 2].tolist())))
# host recomputation of dbias partials from dz (which = 2 is the plain column sum of dz over the 64 rows of the tile)
dz = runs[0]["dz/Conv_2"].reshape(B, 121, 64)
exp = np.stack([dz[(wg % 4) * 64:(wg % 4 + 1) * 64, wg // 4, :].astype(np.float64).sum(0) for wg in range(484)])
if rows == 484:
    err = np.abs(p0[:, 2, :] - exp)
    print("dbias partial vs host sum of dz: max abs err", err.max(), "at", np.unravel_index(err.argmax(), err.shape), "scale", np.abs(exp).max())
