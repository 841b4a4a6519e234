"""Per-leaf gradient error of the HIP path vs the oracle for a tiny net at several batch sizes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "is-dqn_amd"))
import numpy as np, torch
from tests.gpu_helpers import device_batch, make_frame_batch, make_pair
feats = tuple(int(x) for x in os.environ.get("FEATS", "8,8,8,16").split(","))
K, A = int(os.environ.get("K", "3")), int(os.environ.get("A", "4"))
for B in [int(x) for x in os.environ.get("BS", "128,256,257,384,512").split(",")]:
    oracle, eng, params = make_pair(feats, K, A, B, layer_norm=True, seed=7)
    frames, ids, action, reward, terminal, ref = make_frame_batch(B, A, seed=23, n_frames=B + 64)
    batch = device_batch(eng, frames, ids, action, reward, terminal)
    o_grads, _ = oracle.grads(oracle.params, ref)
    with torch.no_grad():
        all_q = oracle.apply(oracle.params, torch.cat((torch.tensor(ref.state), torch.tensor(ref.next_state)))).numpy()
    flat_ids = np.concatenate([ids[:, :4], ids[:, 4:]], 0).copy()
    q = eng.forward(frames=batch._keep[0], frame_stride=frames.shape[1], frame_ids=torch.from_numpy(flat_ids).cuda(), n_rows=2 * B).cpu().numpy().reshape(2 * B, 1 + K, A)
    qerr = np.abs(q - all_q).max()
    o_q, o_t, o_td = oracle.loss_terms(oracle.params, ref)
    grad = torch.zeros_like(eng.params)
    eng.learn_on_batch(batch, grad_out=grad)
    g = eng.internal_to_flax_grads(grad)
    errs = []
    for mod in o_grads:
        for leaf in o_grads[mod]:
            a, b = np.asarray(g[mod][leaf], np.float64), o_grads[mod][leaf].numpy().astype(np.float64)
            errs.append((np.abs(a - b).max() / max(np.abs(b).max(), 1e-12), f"{mod}/{leaf}"))
    errs.sort(reverse=True)
    terr = np.abs(eng.targets.cpu().numpy() - o_t.detach().numpy()).max()
    print(f"B={B}: q err {qerr:.1e} (|q| {np.abs(all_q).max():.2f}) target err {terr:.1e} | " + "  ".join(f"{n} {e:.1e}" for e, n in errs[:6]), flush=True)
