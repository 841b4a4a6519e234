"""ORACLE (test infrastructure) -- n-step frame-stacking replay buffer, CPU restatement.

Follows reference ``slimdqn/sample_collection/replay_buffer.py``:
  * TransitionElement ................... :18-23
  * ReplayElement (no snappy: the ``compress=False`` path the reference tests
    use; pack/unpack are identity round trips here) ... :26-68
  * trajectory accumulator .............. :102-183 (inclusive slice bounds, zero
    padding of young episodes, n-step discounted reward, terminal flush incl.
    the short-episode special case :159-169, truncation clear :180-183)
  * add / FIFO eviction ................. :185-196
  * sample (stack into a batch) ......... :198-213
  * update (forward to the sampler) ..... :215-220

Elements are built here from *window descriptions* (``Window``): for each
element the deque position of the last state frame and of the last next-state
frame (which also fixes the reward span).  The device replay of the product
stores exactly those descriptions (as frame ids), so the same function is the
specification for both.

Pinned by the known answers of the reference's tests/test_replay_buffer.py.
"""
from __future__ import annotations

import collections
import dataclasses
import typing
from typing import Any, Optional

import numpy as np


class TransitionElement(typing.NamedTuple):
    observation: Optional[np.ndarray]
    action: int
    reward: float
    is_terminal: bool
    episode_end: bool = False


@dataclasses.dataclass(frozen=True)
class ReplayElement:
    state: Any
    action: Any
    reward: Any
    next_state: Any
    is_terminal: Any

    def replace(self, **kw):
        return dataclasses.replace(self, **kw)

    # compress=False path: packing is the identity
    def pack(self):
        return self

    def unpack(self):
        return self


class Window(typing.NamedTuple):
    """One replay element described by positions in the trajectory deque.

    The state stack holds deque positions ``state_last-stack+1 .. state_last``
    and the next-state stack ``next_last-stack+1 .. next_last``; positions < 0
    or >= len(deque) are zero frames.  Rewards ``state_last .. next_last-1`` are
    discounted from ``state_last`` (positions >= len(deque) contribute nothing).
    """

    state_last: int
    next_last: int
    is_terminal: bool


class ReplayBuffer:
    def __init__(
        self,
        sampling_distribution,
        batch_size: int,
        max_capacity: int,
        stack_size: int = 4,
        update_horizon: int = 1,
        gamma: float = 0.99,
        checkpoint_duration: int = 4,
        compress: bool = False,
        clipping=None,
    ):
        self.add_count = 0
        self._max_capacity = max_capacity
        self._compress = compress
        self._memory = collections.OrderedDict()
        self._sampling_distribution = sampling_distribution
        self._checkpoint_duration = checkpoint_duration
        self._batch_size = batch_size
        self._stack_size = stack_size
        self._update_horizon = update_horizon
        self._gamma = gamma
        self._clipping = clipping
        self._trajectory = collections.deque(maxlen=update_horizon + stack_size)

    # -- element construction ---------------------------------------------------
    def _build(self, w: Window) -> ReplayElement:
        traj = self._trajectory
        stack, n = self._stack_size, self._update_horizon
        shape = traj[0].observation.shape + (stack,)
        dtype = traj[0].observation.dtype
        state = np.zeros(shape, dtype)
        nxt = np.zeros(shape, dtype)
        reward = 0.0
        for t, tr in enumerate(traj):
            if w.state_last <= t <= w.next_last - 1:
                reward += tr.reward * (self._gamma ** (t - w.state_last))
            s = t - (w.state_last - stack + 1)
            if 0 <= s < stack:
                state[..., s] = tr.observation
            s = t - (w.next_last - stack + 1)
            if 0 <= s < stack:
                nxt[..., s] = tr.observation
        return ReplayElement(
            state=state, action=traj[w.state_last].action, reward=reward, next_state=nxt, is_terminal=w.is_terminal
        )

    def accumulate(self, transition: TransitionElement):
        """Yield the ReplayElements completed by this transition (reference :151-183)."""
        traj = self._trajectory
        stack, n = self._stack_size, self._update_horizon
        traj.append(transition)
        L = len(traj)

        if transition.is_terminal:
            if L < stack + n:
                # terminal before stack+n observations: every not-yet-emitted start
                for state_last in range(max(L - 1 - n, 0), L):
                    next_last = state_last + n
                    yield self._build(Window(state_last, next_last, next_last >= L))
            else:
                # deque is full (L == stack+n): one ordinary element, then the flush
                yield self._build(Window(L - 1 - n, L - 1, False))
                traj.popleft()
                while len(traj) >= stack:
                    yield self._build(Window(stack - 1, stack - 1 + n, True))
                    traj.popleft()
            traj.clear()
        else:
            if L >= 1 + n:
                yield self._build(Window(L - 1 - n, L - 1, False))
            if transition.episode_end:
                traj.clear()

    # -- add / sample / update ----------------------------------------------------
    def add(self, transition: TransitionElement, **kwargs) -> None:
        for element in self.accumulate(transition):
            key = self.add_count
            self._memory[key] = element
            self._sampling_distribution.add(key, **kwargs)
            self.add_count += 1
            if self.add_count > self._max_capacity:
                oldest_key, _ = self._memory.popitem(last=False)
                self._sampling_distribution.remove(oldest_key)

    def sample(self, size=None) -> ReplayElement:
        assert self.add_count, ValueError("No samples in replay buffer!")
        if size is None:
            size = self._batch_size
        keys = self._sampling_distribution.sample(size)
        elems = [self._memory[int(k)] for k in keys]
        return ReplayElement(
            state=np.stack([e.state for e in elems]),
            action=np.stack([e.action for e in elems]),
            reward=np.stack([e.reward for e in elems]),
            next_state=np.stack([e.next_state for e in elems]),
            is_terminal=np.stack([e.is_terminal for e in elems]),
        )

    def update(self, keys, **kwargs) -> None:
        self._sampling_distribution.update(keys, **kwargs)
