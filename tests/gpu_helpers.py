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


def perturbed_params(seed, obs, feats, arch, final_feature, layer_norm, scale=0.1):
    p = onet.init_params(seed, obs, feats, arch, final_feature, layer_norm)
    rng = np.random.default_rng(seed + 1)
    for m in p:
        for n in p[m]:
            if n != "kernel":  # move biases / LN params off their trivial init
                p[m][n] = (p[m][n] + rng.normal(0, scale, p[m][n].shape)).astype(np.float32)
    return p


def make_pair(feats, K, A, B, arch="cnn", obs=(84, 84, 4), layer_norm=True, precision="bf16x3", seed=0,
              lr=1e-3, gamma=0.99, n=1, adam_eps=1.5e-4, dtype=torch.float32):
    from slimdqn._engine import QNetEngine

    params = perturbed_params(seed, obs, feats, arch, (1 + K) * A, layer_norm)
    oracle = OracleAgent(seed, obs, A, K, list(feats), layer_norm, False, arch, lr, gamma, n, 1, 1,
                         adam_eps=adam_eps, dtype=dtype, params=params)
    eng = QNetEngine(obs, A, 1 + K, feats, arch, layer_norm, B, gamma_n=gamma**n, learning_rate=lr,
                     adam_eps=adam_eps, precision=precision)
    eng.import_flax(params)
    return oracle, eng, params


def device_batch(eng, frames, ids, action, reward, terminal):
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    fr, idd, ac, rw, te = d(frames), d(ids), d(action), d(reward), d(terminal)
    batch = eng.make_batch(frames=fr, frame_stride=frames.shape[1], frame_ids=idd, action=ac, reward=rw, terminal=te)
    return batch
