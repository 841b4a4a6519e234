"""Device-resident replay buffer with the reference's surface
(slimdqn/sample_collection/replay_buffer.py:18-220).

What stays on the host (python, like the reference): the n-step trajectory accumulator
(:102-183), FIFO eviction (:185-196) and the sampler's key maps.  What lives in HBM: every
observation ONCE, as a single uint8 frame ``frames[slot][h*w]``, and one table row per replay
element -- the frame slots of its state stack and next-state stack (-1 = the zero frame of the
leading padding, :131-134), action, n-step reward, terminal flag.  The reference stores two full
4-frame stacks per element (snappy-compressed on the host); here 1e6 Atari elements are 7.06 GB of
frames + 41 MB of rows, resident in the 288 GB of one MI355X.

``sample()`` returns a ``DeviceBatch`` with the reference's field names.  The update path consumes
its frame-id table directly (the first convolution gathers pixels through it); ``.state`` /
``.next_state`` materialise the reference's (B, h, w, stack) arrays on demand.
"""
from __future__ import annotations

import collections
import ctypes
import dataclasses
import typing
from collections.abc import Mapping
from typing import Any, Optional

import numpy as np
import torch

from slimdqn import _hip
from slimdqn.sample_collection import ReplayItemID


class TransitionElement(typing.NamedTuple):
    observation: Optional[np.ndarray]
    action: int
    reward: float
    is_terminal: bool
    episode_end: bool = False


@dataclasses.dataclass(frozen=True)
class ReplayElement:
    """replay_buffer.py:26-68.  There is no snappy here: frames are stored once, uncompressed, on the
    device, so pack/unpack are identity round trips."""

    state: Any
    action: Any
    reward: Any
    next_state: Any
    is_terminal: Any

    def replace(self, **kw):
        return dataclasses.replace(self, **kw)

    def pack(self):
        return self

    def unpack(self):
        return self


class DeviceBatch:
    """A sampled batch (ReplayElement fields, replay_buffer.py:198-213) living on the GPU."""

    def __init__(self, rb: "ReplayBuffer", indices, frame_ids, action, reward, is_terminal):
        self._rb = rb
        self.indices = indices  # dense sampler indices (= sum-tree leaves), int32 [B]
        self.frame_ids = frame_ids  # int32 [B][2*stack]
        self.action = action  # int32 [B]
        self.reward = reward  # float32 [B]
        self.is_terminal = is_terminal  # uint8 [B]
        self.frames = rb._frames
        self.frame_stride = rb._hw
        self._stacks = None

    def __len__(self):
        return int(self.action.shape[0])

    def _materialize(self):
        if self._stacks is None:
            rb = self._rb
            B = len(self)
            st = torch.empty(B, rb._h, rb._w, rb._stack_size, dtype=torch.uint8, device=rb.device)
            nx = torch.empty_like(st)
            _hip.check(
                rb._lib.isdqn_replay_materialize(
                    _hip.ptr(self.frames), self.frame_stride, rb._h, rb._w, rb._stack_size, _hip.ptr(self.frame_ids),
                    B, _hip.ptr(st), _hip.ptr(nx), _hip.stream_ptr(rb.device),
                ),
                "isdqn_replay_materialize",
            )
            self._stacks = (st, nx)
        return self._stacks

    def _as_observation(self, stacks):
        """Vector observations were stored as the bytes of their float32 values (ReplayBuffer._alloc_frame): view them back as
        (B, d) rows (stack_size 1: the reference's (d, 1) stacks squeezed, architectures/dqn.py:93) or (B, d, stack)."""
        rb = self._rb
        if not rb._obs_float:
            return stacks
        B = len(self)
        rows = stacks.reshape(B, rb._w, rb._stack_size).permute(0, 2, 1).reshape(B * rb._stack_size, rb._w).contiguous()
        f = rows.view(torch.float32).reshape(B, rb._stack_size, rb._w // 4).permute(0, 2, 1)  # (B, d, stack)
        return f.reshape(B, -1).contiguous() if rb._stack_size == 1 else f.contiguous()

    @property
    def state(self):
        return self._as_observation(self._materialize()[0])

    @property
    def next_state(self):
        return self._as_observation(self._materialize()[1])


class _MemoryView(Mapping):
    """``ReplayBuffer._memory`` as the reference exposes it (an ordered key -> ReplayElement map), backed by
    the device tables; elements are downloaded on access (debug / test use only)."""

    def __init__(self, rb):
        self._rb = rb

    def _range(self):
        rb = self._rb
        return range(max(0, rb.add_count - rb._max_capacity), rb.add_count)

    def __len__(self):
        return len(self._range())

    def __iter__(self):
        return iter(self._range())

    def keys(self):
        return list(self._range())

    def __getitem__(self, key):
        rb = self._rb
        if key not in self._range():
            raise KeyError(key)
        rb._flush()
        slot = key % rb._max_capacity
        ids = rb._h_elem_frames[slot]
        stack = rb._stack_size

        def build(cols):
            out = np.zeros((rb._h, rb._w, stack), np.uint8)
            for c in range(stack):
                if ids[cols + c] >= 0:
                    out[:, :, c] = rb._frames[int(ids[cols + c])].cpu().numpy().reshape(rb._h, rb._w)
            return out

        return ReplayElement(
            state=build(0), action=int(rb._h_elem_action[slot]), reward=float(rb._h_elem_reward64[slot]),
            next_state=build(stack), is_terminal=bool(rb._h_elem_terminal[slot]),
        )


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
        compress: bool = True,
        clipping: callable = None,
        device: str | None = None,
    ):
        self.device = _hip.resolve_device(device)
        if self.device.type == "cuda":
            _hip.require_gpu()
            _hip.bind_device(self.device)
            self._lib = _hip.lib()
        else:
            # host-logic testing only: storage on the CPU, no kernels -- sampling raises (there is no CPU fallback)
            self._lib = None
        self.add_count = 0
        self._max_capacity = max_capacity
        self._compress = compress  # accepted for signature compatibility; frames are stored once, uncompressed
        self._sampling_distribution = sampling_distribution
        self._checkpoint_duration = checkpoint_duration
        self._batch_size = batch_size
        self._stack_size = stack_size
        self._update_horizon = update_horizon
        self._gamma = gamma
        self._clipping = clipping
        self._trajectory = collections.deque()  # entries: (frame_slot, action, reward, is_terminal)
        self._trajectories = {0: self._trajectory}  # one n-step accumulator per transition stream (vectorised envs)
        self._traj_maxlen = update_horizon + stack_size
        self._memory = _MemoryView(self)
        self._frames = None
        self._obs_float = False
        self._h = self._w = self._hw = 0

    # ------------------------------------------------------------------ storage
    def _allocate(self, obs_shape, floating: bool = False):
        # image observations: single 2-D uint8 frames.  Vector observations (LunarLander's (8,) float32, BASELINE configs[0]):
        # the bytes of the float32 vector are stored as one 1 x 4d "frame", so the same tables, kernels and eviction serve both
        self._obs_float = bool(floating)
        if floating:
            assert len(obs_shape) == 1, "floating-point observations must be vectors"
            obs_shape = (1, 4 * int(obs_shape[0]))
        assert len(obs_shape) == 2, "observations must be single 2-D frames (or float vectors)"
        self._h, self._w = int(obs_shape[0]), int(obs_shape[1])
        self._hw = self._h * self._w
        C, s2 = self._max_capacity, 2 * self._stack_size
        self._n_frame_slots = C + self._traj_maxlen + max(64, C // 16)
        self._frames = torch.empty(self._n_frame_slots, self._hw, dtype=torch.uint8, device=self.device)
        self._d_elem_frames = torch.full((C, s2), -1, dtype=torch.int32, device=self.device)
        self._d_elem_action = torch.zeros(C, dtype=torch.int32, device=self.device)
        self._d_elem_reward = torch.zeros(C, dtype=torch.float32, device=self.device)
        self._d_elem_terminal = torch.zeros(C, dtype=torch.uint8, device=self.device)
        self._d_index_to_slot = torch.zeros(C + 1, dtype=torch.int32, device=self.device)
        self._h_elem_frames = np.full((C, s2), -1, np.int32)
        self._h_elem_action = np.zeros(C, np.int32)
        self._h_elem_reward64 = np.zeros(C, np.float64)  # the reference keeps python floats (f64)
        self._h_elem_terminal = np.zeros(C, np.uint8)
        self._h_index_to_slot = np.zeros(C + 1, np.int32)
        self._refcount = np.zeros(self._n_frame_slots, np.int32)
        self._next_fresh = 0
        self._free = []
        self._pending_frames = {}  # slot -> uint8 array (a slot freed and re-used before a flush keeps only the newest)
        self._dirty_rows = []
        self._dirty_index = []

    def _grow_frames(self):
        new_n = int(self._n_frame_slots * 1.25) + 64
        new = torch.empty(new_n, self._hw, dtype=torch.uint8, device=self.device)
        new[: self._n_frame_slots].copy_(self._frames)
        self._frames = new
        self._refcount = np.concatenate([self._refcount, np.zeros(new_n - self._n_frame_slots, np.int32)])
        self._n_frame_slots = new_n

    def _alloc_frame(self, observation) -> int:
        if self._free:
            slot = self._free.pop()
        else:
            if self._next_fresh >= self._n_frame_slots:
                self._flush()
                self._grow_frames()
            slot = self._next_fresh
            self._next_fresh += 1
        obs = np.asarray(observation)
        if self._obs_float:
            obs = np.ascontiguousarray(obs, dtype=np.float32).view(np.uint8)
        elif obs.dtype != np.uint8:
            obs = obs.astype(np.uint8)
        self._pending_frames[slot] = obs.reshape(-1).copy()
        return slot

    def _unref(self, slot: int) -> None:
        if slot < 0:
            return
        self._refcount[slot] -= 1
        if self._refcount[slot] == 0:
            self._free.append(slot)

    def _flush(self) -> None:
        """Upload pending frames / table rows / staged leaf priorities (plumbing copies; the hot path never waits on them)."""
        if self._frames is None:
            return
        flush_sampler = getattr(self._sampling_distribution, "flush", None)
        if flush_sampler is not None:
            flush_sampler()
        if not (self._pending_frames or self._dirty_rows or self._dirty_index):
            return
        # Everything that changed since the last flush goes up in ONE pinned copy and is scattered by ONE launch
        # (isdqn_replay_apply_staged); nine pageable copies and six index_copy_ launches were ~290 us of host time per flush,
        # a quarter of the trainer's loop.
        frames = rows = idx = None
        if self._pending_frames:
            slots = np.fromiter(self._pending_frames.keys(), dtype=np.int32, count=len(self._pending_frames))
            frames = (slots, np.stack(list(self._pending_frames.values())))
            self._pending_frames.clear()
        if self._dirty_rows:
            r = np.unique(np.asarray(self._dirty_rows, dtype=np.int64))
            rows = (r.astype(np.int32), self._h_elem_frames[r], self._h_elem_action[r], self._h_elem_reward64[r].astype(np.float32),
                    self._h_elem_terminal[r])
            self._dirty_rows.clear()
        if self._dirty_index:
            r = np.unique(np.asarray(self._dirty_index, dtype=np.int64))
            idx = (r.astype(np.int32), self._h_index_to_slot[r])
            self._dirty_index.clear()
        if self.device.type != "cuda":  # host-logic tests: storage on the CPU, plain indexed writes
            if frames is not None:
                self._frames[torch.from_numpy(frames[0].astype(np.int64))] = torch.from_numpy(frames[1])
            if rows is not None:
                at = torch.from_numpy(rows[0].astype(np.int64))
                for dst, v in zip((self._d_elem_frames, self._d_elem_action, self._d_elem_reward, self._d_elem_terminal), rows[1:]):
                    dst[at] = torch.from_numpy(np.ascontiguousarray(v))
            if idx is not None:
                self._d_index_to_slot[torch.from_numpy(idx[0].astype(np.int64))] = torch.from_numpy(idx[1])
            return
        u = _hip.StagedUpdates()
        sections = []  # (field name of the offset, array)
        if frames is not None:
            u.n_frames, u.frame_bytes = len(frames[0]), self._hw
            sections += [("off_frame_slots", frames[0]), ("off_frame_data", frames[1])]
        u.stack2 = 2 * self._stack_size
        if rows is not None:
            u.n_rows = len(rows[0])
            sections += list(zip(("off_rows", "off_row_frames", "off_row_action", "off_row_reward", "off_row_terminal"), rows))
        if idx is not None:
            u.n_index = len(idx[0])
            sections += [("off_index_rows", idx[0]), ("off_index_vals", idx[1])]
        align = lambda n: (n + 15) & ~15
        total = sum(align(a.nbytes) for _n, a in sections)
        st = getattr(self, "_stage", None)
        if st is None or st["host"].numel() < total:
            n = max(total * 2, 1 << 20)
            st = self._stage = dict(host=torch.empty(n, dtype=torch.uint8).pin_memory(),
                                    dev=torch.empty(n, dtype=torch.uint8, device=self.device), done=torch.cuda.Event())
            st["np"] = st["host"].numpy()
        else:
            st["done"].synchronize()  # the previous flush's copy has left the pinned buffer (long ago)
        off = 0
        for name, a in sections:
            a = np.ascontiguousarray(a)
            st["np"][off : off + a.nbytes] = a.reshape(-1).view(np.uint8)
            setattr(u, name, off)
            off += align(a.nbytes)
        st["dev"][:off].copy_(st["host"][:off], non_blocking=True)
        st["done"].record()
        _hip.check(
            self._lib.isdqn_replay_apply_staged(
                _hip.ptr(st["dev"]), ctypes.byref(u), _hip.ptr(self._frames), self._hw, _hip.ptr(self._d_elem_frames),
                _hip.ptr(self._d_elem_action), _hip.ptr(self._d_elem_reward), _hip.ptr(self._d_elem_terminal),
                _hip.ptr(self._d_index_to_slot), _hip.stream_ptr(self.device),
            ),
            "isdqn_replay_apply_staged",
        )

    # ------------------------------------------------------------------ trajectory accumulator (:102-183)
    def _window_ids(self, last: int):
        traj, stack = self._trajectory, self._stack_size
        L = len(traj)
        return [traj[p][0] if 0 <= p < L else -1 for p in range(last - stack + 1, last + 1)]

    def _element(self, state_last: int, next_last: int, is_terminal: bool):
        traj = self._trajectory
        reward = 0.0
        for t in range(state_last, min(next_last, len(traj))):  # rewards state_last .. next_last-1 (:138-143)
            reward += traj[t][2] * (self._gamma ** (t - state_last))
        return (self._window_ids(state_last) + self._window_ids(next_last), traj[state_last][1], reward, is_terminal)

    def _pop_left(self):
        slot = self._trajectory.popleft()[0]
        self._unref(slot)

    def _clear_trajectory(self):
        while self._trajectory:
            self._pop_left()

    def accumulate(self, transition: TransitionElement):
        """Yield (frame_ids[2*stack], action, reward, is_terminal) for every element this transition completes."""
        if self._frames is None:
            first = np.asarray(transition.observation)
            self._allocate(first.shape, floating=first.ndim == 1 and first.dtype.kind == "f")
        stack, n = self._stack_size, self._update_horizon
        if len(self._trajectory) == self._traj_maxlen:  # deque(maxlen=...) semantics of the reference (:100)
            self._pop_left()
        slot = self._alloc_frame(transition.observation)
        self._refcount[slot] += 1
        self._trajectory.append((slot, transition.action, transition.reward, transition.is_terminal))
        L = len(self._trajectory)

        if transition.is_terminal:
            if L < stack + n:  # terminal before stack+n observations (:159-169)
                for state_last in range(max(L - 1 - n, 0), L):
                    next_last = state_last + n
                    yield self._element(state_last, next_last, next_last >= L)
            else:  # (:170-176)
                yield self._element(L - 1 - n, L - 1, False)
                self._pop_left()
                while len(self._trajectory) >= stack:
                    yield self._element(stack - 1, stack - 1 + n, True)
                    self._pop_left()
            self._clear_trajectory()
        else:
            if L >= 1 + n:
                yield self._element(L - 1 - n, L - 1, False)
            if transition.episode_end:
                self._clear_trajectory()

    # ------------------------------------------------------------------ add / sample / update (:185-220)
    def _mark_index(self, index: int, key: int) -> None:
        self._h_index_to_slot[index] = key % self._max_capacity
        self._dirty_index.append(index)

    def add(self, transition: TransitionElement, stream: int = 0, **kwargs: Any) -> None:
        """replay_buffer.py:185-196.  ``stream``: which environment of a vectorised collector the transition comes from --
        each stream has its own trajectory accumulator (frame stacks and n-step returns never mix environments), all of them
        feed the same element table in arrival order."""
        if stream not in self._trajectories:
            self._trajectories[stream] = collections.deque()
        self._trajectory = self._trajectories[stream]
        sampler = self._sampling_distribution
        C = self._max_capacity
        for ids, action, reward, is_terminal in self.accumulate(transition):
            key = ReplayItemID(self.add_count)
            slot = key % C
            evict = self.add_count >= C  # the row being overwritten belongs to key - C, evicted below
            old_ids = self._h_elem_frames[slot].copy() if evict else None
            for f in ids:
                if f >= 0:
                    self._refcount[f] += 1
            self._h_elem_frames[slot] = ids
            self._h_elem_action[slot] = action
            self._h_elem_reward64[slot] = reward
            self._h_elem_terminal[slot] = 1 if is_terminal else 0
            self._dirty_rows.append(slot)
            sampler.add(key, **kwargs)
            self._mark_index(sampler._key_to_index[key], key)
            self.add_count += 1
            if self.add_count > C:
                oldest_key = self.add_count - 1 - C
                hole = sampler._key_to_index[oldest_key]
                sampler.remove(oldest_key)
                if hole < len(sampler._index_to_key):
                    self._mark_index(hole, sampler._index_to_key[hole])
                for f in old_ids:
                    self._unref(int(f))

    def sample(self, size=None) -> DeviceBatch:
        assert self.add_count, ValueError("No samples in replay buffer!")
        if size is None:
            size = self._batch_size
        self._flush()
        indices = self._sampling_distribution.sample_device(size)
        return self.gather(indices)

    def gather(self, indices: torch.Tensor) -> DeviceBatch:
        """Rows of the element table for dense sampler indices (device int32)."""
        if self._lib is None:
            raise RuntimeError("ReplayBuffer.sample needs the HIP path (device='cuda:N'); there is no CPU fallback")
        B = int(indices.numel())
        s2 = 2 * self._stack_size
        ids = torch.empty(B, s2, dtype=torch.int32, device=self.device)
        action = torch.empty(B, dtype=torch.int32, device=self.device)
        reward = torch.empty(B, dtype=torch.float32, device=self.device)
        terminal = torch.empty(B, dtype=torch.uint8, device=self.device)
        _hip.check(
            self._lib.isdqn_replay_gather_rows(
                _hip.ptr(self._d_elem_frames), _hip.ptr(self._d_elem_action), _hip.ptr(self._d_elem_reward),
                _hip.ptr(self._d_elem_terminal), self._stack_size, _hip.ptr(self._d_index_to_slot), _hip.ptr(indices),
                B, _hip.ptr(ids), _hip.ptr(action), _hip.ptr(reward), _hip.ptr(terminal), _hip.stream_ptr(self.device),
            ),
            "isdqn_replay_gather_rows",
        )
        return DeviceBatch(self, indices, ids, action, reward, terminal)

    def update(self, keys, **kwargs: Any) -> None:
        self._sampling_distribution.update(keys, **kwargs)

    def update_device(self, batch: DeviceBatch, priorities: torch.Tensor) -> None:
        """Priority writeback for a sampled batch without leaving the GPU (north star: TD-error writeback)."""
        self._sampling_distribution.update_device(batch.indices, priorities)

    # ------------------------------------------------------------------ synthetic prefill (benchmarks)
    def prefill_synthetic(self, n_elements: int, obs_shape=(84, 84), n_actions: int = 9, seed: int = 0,
                          p_terminal: float = 0.005, priorities=None) -> None:
        """Fill the buffer with ``n_elements`` elements of one long synthetic stream: element i has state
        frames i..i+stack-1 and next-state frames i+n..i+n+stack-1 (uniform random uint8 pixels, i.e.
        incompressible), random actions, rewards in {-1,0,1} (p = .05/.9/.05) and Bernoulli terminals."""
        assert self.add_count == 0 and n_elements <= self._max_capacity
        if self._frames is None:
            self._allocate(obs_shape)
        stack, n = self._stack_size, self._update_horizon
        n_frames = n_elements + stack + n - 1
        assert n_frames <= self._n_frame_slots
        g = torch.Generator(device=self.device).manual_seed(seed)
        chunk = 65536
        for s in range(0, n_frames, chunk):
            e = min(n_frames, s + chunk)
            self._frames[s:e] = torch.randint(0, 256, (e - s, self._hw), dtype=torch.uint8, device=self.device, generator=g)
        rng = np.random.default_rng(seed)
        base = np.arange(n_elements, dtype=np.int32)[:, None]
        self._h_elem_frames[:n_elements, :stack] = base + np.arange(stack, dtype=np.int32)[None]
        self._h_elem_frames[:n_elements, stack:] = base + n + np.arange(stack, dtype=np.int32)[None]
        self._h_elem_action[:n_elements] = rng.integers(0, n_actions, n_elements)
        self._h_elem_reward64[:n_elements] = rng.choice([-1.0, 0.0, 1.0], size=n_elements, p=[0.05, 0.9, 0.05])
        self._h_elem_terminal[:n_elements] = rng.random(n_elements) < p_terminal
        self._h_index_to_slot[:n_elements] = np.arange(n_elements, dtype=np.int32)
        np.add.at(self._refcount, self._h_elem_frames[:n_elements].reshape(-1), 1)
        self._next_fresh = n_frames
        self._d_elem_frames.copy_(torch.from_numpy(self._h_elem_frames))
        self._d_elem_action.copy_(torch.from_numpy(self._h_elem_action))
        self._d_elem_reward.copy_(torch.from_numpy(self._h_elem_reward64.astype(np.float32)))
        self._d_elem_terminal.copy_(torch.from_numpy(self._h_elem_terminal))
        self._d_index_to_slot.copy_(torch.from_numpy(self._h_index_to_slot))
        self.add_count = n_elements
        self._sampling_distribution.add_bulk(range(n_elements), priorities)
