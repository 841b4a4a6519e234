"""Host-side ALE environment exposing the surface the trainer consumes (env row):
``state`` (float32 (84,84,4)), ``observation`` (uint8 (84,84), newest frame), ``n_actions``, ``n_steps``,
``state_height`` / ``state_width`` / ``n_stacked_frames``, ``reset()`` and ``step(a) -> (reward, terminal)``.

Pre-processing follows the description of slimdqn/environments/atari.py:13-89 (Nature-DQN protocol): one
agent step = 4 emulator frames with the action repeated, the observation is the pixel-wise maximum of the
last two grayscale screens, area-resized to 84x84; sticky actions with probability 0.25; minimal action set.
It stays on host cores (north star) -- nothing here touches the GPU.  gymnasium / ale_py / cv2 are not part
of the build image; they are imported lazily so that this module always imports.
"""
from __future__ import annotations

import collections

import numpy as np

FRAME_SIDE = 84
STACK = 4
ACTION_REPEAT = 4
STICKY_PROBABILITY = 0.25


class _Screens:
    """Two grayscale screen buffers of the emulator's native size and their pooled, resized view."""

    def __init__(self, ale, height: int, width: int) -> None:
        self._ale = ale
        self._pair = np.zeros((2, height, width), dtype=np.uint8)

    def grab(self, which: int) -> None:
        self._ale.getScreenGrayscale(self._pair[which])

    def clear(self, which: int) -> None:
        self._pair[which] = 0

    def frame(self, pooled: bool) -> np.ndarray:
        import cv2

        src = self._pair.max(axis=0) if pooled else self._pair[0]
        small = cv2.resize(src, (FRAME_SIDE, FRAME_SIDE), interpolation=cv2.INTER_AREA)
        return small.astype(np.uint8, copy=False)


class AtariEnv:
    state_height = FRAME_SIDE
    state_width = FRAME_SIDE
    n_stacked_frames = STACK
    n_skipped_frames = ACTION_REPEAT

    def __init__(self, name: str) -> None:
        import ale_py  # noqa: F401  registers the "ALE/" namespace with gymnasium
        import gymnasium

        self.name = name
        made = gymnasium.make(
            f"ALE/{name}-v5",
            frameskip=1,
            repeat_action_probability=STICKY_PROBABILITY,
            full_action_space=False,
            max_num_frames_per_episode=100_000,
        )
        self.env = made.unwrapped
        self.n_actions = int(self.env.action_space.n)
        native_h, native_w = self.env.observation_space.shape[:2]
        self._screens = _Screens(self.env.ale, native_h, native_w)
        self._stack = collections.deque(maxlen=STACK)
        self.n_steps = 0

    # -- views the trainer reads -----------------------------------------------------------------
    @property
    def state(self) -> np.ndarray:
        return np.stack(self._stack, axis=-1).astype(np.float32)

    @property
    def observation(self) -> np.ndarray:
        return self._stack[-1].copy()

    # -- dynamics ----------------------------------------------------------------------------------
    def reset(self) -> None:
        self.env.reset()
        self.n_steps = 0
        self._screens.grab(0)
        self._screens.clear(1)
        blank = np.zeros((FRAME_SIDE, FRAME_SIDE), dtype=np.uint8)
        self._stack.clear()
        self._stack.extend([blank] * (STACK - 1))
        self._stack.append(self._screens.frame(pooled=False))

    def step(self, action: int):
        total, terminal = 0.0, False
        for repeat in range(ACTION_REPEAT):
            _, r, terminal, _, _ = self.env.step(action)
            total += r
            remaining = ACTION_REPEAT - 1 - repeat
            if remaining < 2:  # keep the last two screens for the max-pool
                self._screens.grab(1 - remaining)
            if terminal:
                break
        self._stack.append(self._screens.frame(pooled=True))
        self.n_steps += 1
        return total, terminal
