"""Host-visible latency of the acting path at the c2 network (A=9, K=9): best_action per environment step (frame ring + weight
mirror reuse) and best_actions for n environments; the legacy full-upload entry for comparison."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "is-dqn_amd"))
from slimdqn.environments.synthetic import SyntheticAtariEnv
from slimdqn.networks.isdqn import iSDQN

agent = iSDQN(0, (84, 84, 4), 9, 9, [32, 64, 64, 512], True, False, "cnn", 6.25e-5, 0.99, 1, 4, 8000, adam_eps=1.5e-4, batch_size=256)
env = SyntheticAtariEnv("Synthetic", seed=0, episode_length=10**9)
env.reset()
out = {}


def timed(fn, n):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e6


def step_ring():
    agent.best_action(agent.params, env.state, key=3)
    env.step(0)


def step_env_only():
    _ = env.state
    env.step(0)


def step_legacy():
    eng = agent._engine
    s = np.asarray(env.state)
    planes = np.ascontiguousarray(np.moveaxis(s, -1, 0)).reshape(4, 84 * 84).astype(np.uint8)
    fr = torch.from_numpy(planes).to(eng.device)
    ids = torch.arange(4, dtype=torch.int32, device=eng.device)
    int(eng.best_action(idx_network=3, frames=fr, frame_stride=84 * 84, frame_ids=ids).item())
    env.step(0)


out["env_only_us"] = timed(step_env_only, 500)
out["best_action_legacy_us"] = timed(step_legacy, 500) - out["env_only_us"]
out["best_action_ring_mirror_us"] = timed(step_ring, 500) - out["env_only_us"]
for n in (8, 32, 128, 512):
    states = np.random.default_rng(0).integers(0, 256, (n, 84, 84, 4), dtype=np.uint8)
    out[f"best_actions_n{n}_us_per_action"] = timed(lambda: agent.best_actions(agent.params, states, key=3), 100) / n
print(json.dumps(out))
