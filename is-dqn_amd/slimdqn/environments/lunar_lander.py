"""LunarLander on the host cores with the reference's surface (slimdqn/environments/lunar_lander.py:5-23): ``state`` /
``observation`` (float32 (8,)), ``observation_shape``, ``n_actions``, ``n_steps``, ``reset()``, ``step(a) -> (reward, absorbing)``.
BASELINE configs[0] (LunarLander iS-DQN K=1, MLP head, uniform replay, batch 32).  gymnasium / Box2D are not part of the build
image: they are imported lazily, and ``SyntheticLunarLander`` stands in for them offline (same surface, seeded toy dynamics)."""
from __future__ import annotations

import numpy as np


class LunarLander:
    def __init__(self, render_mode=None):
        import gymnasium as gym

        self.env = gym.make("LunarLander-v3", render_mode=render_mode)
        self.observation_shape = self.env.observation_space.shape
        self.n_actions = int(self.env.action_space.n)
        self.n_steps = 0

    @property
    def observation(self) -> np.ndarray:
        return np.copy(self.state)

    def reset(self):
        self.state, _ = self.env.reset()
        self.n_steps = 0

    def step(self, action):
        self.state, reward, absorbing, _, _ = self.env.step(action)
        self.n_steps += 1
        return reward, absorbing


class SyntheticLunarLander:
    """Eight float32 observation components, four actions: a damped point mass with thrust noise that "lands" (absorbing,
    reward by distance) when its height reaches zero.  Not Box2D physics -- a seeded stand-in with the environment's shapes,
    dtypes and episode structure so that the fc trainer runs end to end where gymnasium is absent."""

    observation_shape = (8,)
    n_actions = 4

    def __init__(self, seed: int = 0, episode_length: int = 120):
        self._rng = np.random.default_rng(seed)
        self._episode_length = episode_length
        self.n_steps = 0
        self.state = np.zeros(8, np.float32)

    @property
    def observation(self) -> np.ndarray:
        return np.copy(self.state)

    def reset(self):
        s = np.zeros(8, np.float32)
        s[0] = self._rng.uniform(-0.3, 0.3)   # x
        s[1] = self._rng.uniform(1.2, 1.5)    # height
        s[2:4] = self._rng.normal(0, 0.1, 2)  # velocities
        s[4] = self._rng.normal(0, 0.05)      # angle
        self.state = s
        self.n_steps = 0

    def step(self, action):
        s = self.state.astype(np.float64)
        thrust = {0: (0.0, 0.0), 1: (-0.03, 0.0), 2: (0.0, 0.06), 3: (0.03, 0.0)}[int(action)]
        s[2] = 0.98 * s[2] + thrust[0] + self._rng.normal(0, 0.005)
        s[3] = 0.98 * s[3] - 0.03 + thrust[1] + self._rng.normal(0, 0.005)
        s[0] += s[2] * 0.1
        s[1] += s[3] * 0.1
        s[5] = 0.9 * s[5] + 0.1 * (thrust[0] * 5)
        s[4] += s[5] * 0.1
        landed = s[1] <= 0.0
        self.n_steps += 1
        absorbing = bool(landed or self.n_steps >= self._episode_length)
        if landed:
            s[1] = 0.0
            s[6] = s[7] = 1.0
        reward = float(-0.3 * (action != 0) - abs(s[0]) * 0.1 + (100.0 * (1.0 - min(1.0, abs(s[0]) + abs(s[3]))) if landed else 0.0))
        self.state = s.astype(np.float32)
        return reward, absorbing
