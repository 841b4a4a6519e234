"""A deterministic stand-in for AtariEnv (same surface: state / observation / n_actions / n_steps / reset /
step) producing random 84x84 uint8 frames.  ALE is not installed in the build image; this lets the trainer
loop, the device replay and the update path run end to end on the GPU box (smoke tests, host-side timing)."""
from __future__ import annotations

import numpy as np


class SyntheticAtariEnv:
    def __init__(self, name: str = "Synthetic", n_actions: int = 9, seed: int = 0, episode_length: int = 200) -> None:
        self.name = name
        self.state_height, self.state_width = (84, 84)
        self.n_stacked_frames = 4
        self.n_actions = n_actions
        self._rng = np.random.default_rng(seed)
        self._episode_length = episode_length
        self.n_steps = 0
        self.state_ = np.zeros((84, 84, 4), dtype=np.uint8)

    @property
    def state(self) -> np.ndarray:
        return np.array(self.state_, dtype=np.float32)

    @property
    def observation(self) -> np.ndarray:
        return np.copy(self.state_[:, :, -1])

    def _frame(self):
        return self._rng.integers(0, 256, size=(84, 84), dtype=np.uint8)

    def reset(self):
        self.n_steps = 0
        self.state_ = np.zeros((84, 84, 4), dtype=np.uint8)
        self.state_[:, :, -1] = self._frame()

    def step(self, action):
        self.state_ = np.roll(self.state_, -1, axis=-1)
        self.state_[:, :, -1] = self._frame()
        self.n_steps += 1
        reward = float(self._rng.integers(-1, 2))
        terminal = self.n_steps >= self._episode_length
        return reward, terminal
