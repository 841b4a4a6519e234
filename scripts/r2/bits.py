"""Bit fingerprint of a few learn steps (params, Adam moments, losses, q / targets / priorities) for several network shapes:
two builds of the library must print identical lines when a change is meant to be arithmetic-neutral (ISDQN_HIP_LIB)."""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "is-dqn_amd"))
import numpy as np, torch
from tests.gpu_helpers import make_frame_batch, device_batch
from slimdqn._engine import QNetEngine

def h(t):
    return hashlib.sha256(t.detach().cpu().numpy().tobytes()).hexdigest()[:12]

CASES = [((32, 64, 64, 512), 9, 9, 256, True, (84, 84, 4), "bf16x3"), ((32, 64, 64, 512), 32, 4, 1024, True, (84, 84, 4), "bf16x3"),
         ((7, 9, 11, 13), 3, 5, 6, True, (84, 84, 4), "bf16x3"), ((16, 20, 5, 24), 2, 3, 5, False, (84, 84, 4), "bf16x3"),
         ((32, 64, 64, 512), 9, 9, 33, True, (84, 84, 4), "bf16"), ((7, 9, 11, 13), 2, 3, 5, True, (44, 44, 6), "bf16x3"),
         ((16, 20, 12, 24), 3, 4, 6, True, (52, 60, 2), "bf16x3"), ((8, 8, 8, 16), 0, 4, 64, True, (84, 84, 4), "bf16x3")]
for feats, K, A, B, ln, obs, prec in CASES:
    frames, ids, action, reward, terminal, _ = make_frame_batch(B, A, seed=5, h=obs[0], w=obs[1], stack=obs[2])
    eng = QNetEngine(obs, A, 1 + K if K else 1, feats, "cnn", ln, B, gamma_n=0.99, learning_rate=1e-3, adam_eps=1.5e-4, precision=prec)
    eng.init_params(1)
    b = device_batch(eng, frames, ids, action, reward, terminal)
    for _ in range(3):
        eng.learn_on_batch(b)
    pre = eng.loss_on_batch(b).clone()
    fr = torch.from_numpy(frames).cuda()
    one = torch.from_numpy(ids[:1, : obs[2]].copy()).cuda()
    q1 = eng.forward(frames=fr, frame_stride=frames.shape[1], frame_ids=one, n_rows=1)
    torch.cuda.synchronize()
    n = eng.n_param_floats
    print(f"{feats} K={K} A={A} B={B} ln={ln} obs={obs} {prec}: params {h(eng.params[:n])} m {h(eng.adam_m[:n])} v {h(eng.adam_v[:n])} losses {h(eng.losses)} "
          f"acc {h(eng.losses_accum)} q {h(eng.q_values)} t {h(eng.targets)} pri {h(eng.priorities)} loss_only {h(pre)} fwd1 {h(q1)}")
# fc architecture
from slimdqn._engine import QNetEngine as E
rng = np.random.default_rng(0)
for feats, K, A, B in (((100, 100), 1, 4, 32), ((300, 600), 3, 5, 17)):
    eng = E((8,), A, 1 + K, feats, "fc", True, B, gamma_n=0.99, learning_rate=3e-4, adam_eps=1e-8)
    eng.init_params(2)
    d = lambda a: torch.from_numpy(a).cuda()
    b = eng.make_batch(state=d(rng.normal(size=(B, 8)).astype(np.float32)), next_state=d(rng.normal(size=(B, 8)).astype(np.float32)),
                       action=d(rng.integers(0, A, B).astype(np.int32)), reward=d(rng.normal(size=B).astype(np.float32)), terminal=d((rng.random(B) < 0.2).astype(np.uint8)))
    for _ in range(3):
        eng.learn_on_batch(b)
    torch.cuda.synchronize()
    n = eng.n_param_floats
    print(f"fc {feats} K={K} B={B}: params {h(eng.params[:n])} m {h(eng.adam_m[:n])} losses {h(eng.losses)} q {h(eng.q_values)}")
