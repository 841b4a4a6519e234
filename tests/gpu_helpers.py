"""Shared helpers for the GPU parity tests: synthetic replay batches in both layouts (single frames +
id table for the HIP path, channel-last stacks for the oracle) and oracle/engine pairs that start
from identical parameters."""
import numpy as np
import torch

from oracle import network as onet
from oracle.isdqn import iSDQN as OracleAgent
from oracle.replay_buffer import ReplayElement


def make_frame_batch(B, n_actions, seed=0, h=84, w=84, stack=4, n_frames=None, zero_frac=0.1):
    """Random single frames + id table [B][2*stack] (-1 = zero frame) and the equivalent
    reference-layout batch (state/next_state (B,h,w,stack) uint8)."""
    rng = np.random.default_rng(seed)
    n_frames = n_frames or (3 * B + 8)
    frames = rng.integers(0, 256, size=(n_frames, h * w), dtype=np.uint8)
    ids = rng.integers(0, n_frames, size=(B, 2 * stack)).astype(np.int32)
    ids[rng.random(ids.shape) < zero_frac] = -1
    action = rng.integers(0, n_actions, B).astype(np.int32)
    reward = rng.normal(size=B).astype(np.float32)
    terminal = (rng.random(B) < 0.3).astype(np.uint8)

    def stacks(cols):
        out = np.zeros((B, h, w, stack), np.uint8)
        for b in range(B):
            for c in range(stack):
                if ids[b, cols + c] >= 0:
                    out[b, :, :, c] = frames[ids[b, cols + c]].reshape(h, w)
        return out

    ref = ReplayElement(state=stacks(0), action=action.astype(np.int64), reward=reward.astype(np.float64),
                        next_state=stacks(stack), is_terminal=terminal.astype(np.int64))
    return frames, ids, action, reward, terminal, ref


def perturbed_params(seed, obs, feats, arch, final_feature, layer_norm, scale=0.1, batch_norm=False):
    p = onet.init_params(seed, obs, feats, arch, final_feature, layer_norm, batch_norm=batch_norm)
    rng = np.random.default_rng(seed + 1)
    for m in p:
        for n in p[m]:
            if n != "kernel":  # move biases / LN params off their trivial init
                p[m][n] = (p[m][n] + rng.normal(0, scale, p[m][n].shape)).astype(np.float32)
    return p


def make_pair(feats, K, A, B, arch="cnn", obs=(84, 84, 4), layer_norm=True, precision="bf16x3", seed=0,
              lr=1e-3, gamma=0.99, n=1, adam_eps=1.5e-4, dtype=torch.float32, batch_norm=False):
    """``batch_norm``: both sides also start from the same (non-trivial) running averages."""
    from slimdqn._engine import QNetEngine

    params = perturbed_params(seed, obs, feats, arch, (1 + K) * A, layer_norm, batch_norm=batch_norm)
    oracle = OracleAgent(seed, obs, A, K, list(feats), layer_norm, batch_norm, arch, lr, gamma, n, 1, 1,
                         adam_eps=adam_eps, dtype=dtype, params=params)
    eng = QNetEngine(obs, A, 1 + K, feats, arch, layer_norm, B, gamma_n=gamma**n, learning_rate=lr,
                     adam_eps=adam_eps, precision=precision, batch_norm=batch_norm)
    stats = None
    if batch_norm:
        rng = np.random.default_rng(seed + 2)
        stats = {m: {"mean": rng.normal(0, 0.3, l["mean"].shape).astype(np.float32), "var": rng.uniform(0.5, 2.0, l["var"].shape).astype(np.float32)}
                 for m, l in onet.init_batch_stats(params).items()}
        oracle.batch_stats = onet.to_torch(stats, dtype)
    eng.import_flax(params, batch_stats=stats)
    return oracle, eng, params


def device_batch(eng, frames, ids, action, reward, terminal):
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    fr, idd, ac, rw, te = d(frames), d(ids), d(action), d(reward), d(terminal)
    batch = eng.make_batch(frames=fr, frame_stride=frames.shape[1], frame_ids=idd, action=ac, reward=rw, terminal=te)
    return batch


def masked_reference_grads(params, feats, K, A, ref, z_hip, layer_norm=True, gamma_n=0.99, huber_delta=0.0):
    """float64 torch gradients of the iS-DQN loss (isdqn.py:92-109) in which every ReLU of the ONLINE half takes its
    pass/block decision from the HIP path's own pre-activations `z_hip[layer]` ([B, ...] float32, layer names Conv_0..2,
    Dense_0).  Why: a batch holds 10^5..10^7 ReLU inputs, a few of them within the forward's 1e-5 of zero; where the HIP
    mask differs from an independent forward's, the whole upstream gradient of that image changes by O(1), i.e. the leaves
    by O(1/B) -- as much as a missing image would.  With the decisions pinned, what is left is arithmetic, and the
    gradient comparison can be held to 1e-4 instead of 3e-3.  Returns a Flax-layout pytree of float64 arrays."""
    import torch.nn.functional as F

    B = len(ref.action)
    P = {m: {k: torch.tensor(np.asarray(v), dtype=torch.float64, requires_grad=True) for k, v in d.items()} for m, d in params.items()}

    def same(size, k, s):
        out = -(-size // s)
        tot = max((out - 1) * s + k - size, 0)
        return tot // 2, tot - tot // 2

    def ln(z, name):
        if not layer_norm:
            return z
        mean = z.mean(-1, keepdim=True)
        var = ((z * z).mean(-1, keepdim=True) - mean * mean).clamp_min(0)
        return (z - mean) * torch.rsqrt(var + 1e-6) * P[name]["scale"] + P[name]["bias"]

    def relu_with_hip_mask(y, z_name, ln_name):
        # online rows: decision from the HIP pre-activations; next-state rows (no gradient flows there): plain ReLU
        zh = torch.tensor(np.asarray(z_hip[z_name], np.float64)).reshape((B,) + tuple(y.shape[1:]))
        with torch.no_grad():
            keep_online = (ln(zh, ln_name) > 0).to(torch.float64)
        return torch.cat([y[:B] * keep_online, torch.relu(y[B:])])

    x = torch.tensor(np.concatenate([ref.state, ref.next_state]), dtype=torch.float64) / 255.0  # NHWC
    n_ln = 0
    for i, (k, s) in enumerate(((8, 4), (4, 2), (3, 1))):
        lo_h, hi_h = same(x.shape[1], k, s)
        lo_w, hi_w = same(x.shape[2], k, s)
        xp = F.pad(x.permute(0, 3, 1, 2), (lo_w, hi_w, lo_h, hi_h))
        z = F.conv2d(xp, P[f"Conv_{i}"]["kernel"].permute(3, 2, 0, 1), P[f"Conv_{i}"]["bias"], stride=s).permute(0, 2, 3, 1)
        x = relu_with_hip_mask(ln(z, f"LayerNorm_{n_ln}"), f"Conv_{i}", f"LayerNorm_{n_ln}")
        n_ln += 1
    h = x.reshape(2 * B, -1)
    n_dense = 0
    for width in feats[3:]:
        z = h @ P[f"Dense_{n_dense}"]["kernel"] + P[f"Dense_{n_dense}"]["bias"]
        h = relu_with_hip_mask(ln(z, f"LayerNorm_{n_ln}"), f"Dense_{n_dense}", f"LayerNorm_{n_ln}")
        n_ln += 1
        n_dense += 1
    q = (h @ P[f"Dense_{n_dense}"]["kernel"] + P[f"Dense_{n_dense}"]["bias"]).reshape(2 * B, 1 + K, A)
    act = torch.tensor(np.asarray(ref.action), dtype=torch.long)
    qv = q[:B, 1:, :][torch.arange(B), :, act]
    r = torch.tensor(np.asarray(ref.reward, np.float64))
    t = torch.tensor(np.asarray(ref.is_terminal, np.float64))
    tg = r[:, None] + (1 - t)[:, None] * gamma_n * q[B:, :K].max(-1).values
    d = qv - tg.detach()
    if huber_delta > 0:
        td = torch.where(d.abs() <= huber_delta, 0.5 * d * d, huber_delta * (d.abs() - 0.5 * huber_delta))
    else:
        td = d**2
    loss = td.mean(0).sum()
    loss.backward()
    return {m: {k: v.grad.numpy() for k, v in d.items()} for m, d in P.items()}
