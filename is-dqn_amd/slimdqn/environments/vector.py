"""n host environments stepped in lockstep (not in the reference, whose loop drives one ALE instance one action at a time:
slimdqn/sample_collection/utils.py:21-43).  The emulators stay on the host cores; what is batched is the device side of
acting: one ``best_actions`` forward and one device->host read per round of n environment steps instead of n of each."""
from __future__ import annotations

import numpy as np


class VectorEnv:
    def __init__(self, envs):
        self.envs = list(envs)
        assert len(self.envs) >= 1
        e = self.envs[0]
        self.n_actions = e.n_actions
        self.state_height, self.state_width, self.n_stacked_frames = e.state_height, e.state_width, e.n_stacked_frames

    def __len__(self) -> int:
        return len(self.envs)

    def reset(self) -> None:
        for e in self.envs:
            e.reset()

    @property
    def states(self) -> np.ndarray:
        """(n, h, w, stack) uint8: the frame stacks every environment would hand to ``best_action``."""
        return np.stack([np.asarray(e.state).astype(np.uint8, copy=False) for e in self.envs])
