"""Engine plumbing shared by the agents (iSDQN, DQN, TFDQN): one QNetEngine per batch size, parameter handles, batch
conversion, single-observation forward.  No arithmetic of the hot path lives here."""
from __future__ import annotations

import numpy as np
import torch

from slimdqn import _hip
from slimdqn._engine import QNetEngine


class DeviceParams:
    """Handle on the flat device parameter buffer; ``handle["params"]`` gives the Flax-layout pytree (a copy)."""

    def __init__(self, engine: QNetEngine, tensor: torch.Tensor):
        self._engine = engine
        self.tensor = tensor

    def to_flax(self):
        return self._engine.export_flax(self.tensor)

    def __getitem__(self, key):
        if key != "params":
            raise KeyError(key)
        return self.to_flax()

    def clone(self) -> "DeviceParams":
        return DeviceParams(self._engine, self.tensor.clone())

    copy = clone  # the reference's ``self.params.copy()`` (dqn.py:34, 50)


class EngineAgent:
    """Common state: ``n_heads`` network heads of ``n_actions`` outputs each on the HIP engine."""

    def _init_engine_agent(self, key, observation_dim, n_actions, n_heads, features, layer_norm, architecture_type,
                           learning_rate, gamma, update_horizon, adam_eps, batch_size, precision, device, huber_delta=0.0):
        self.n_actions = n_actions
        self._n_heads = int(n_heads)
        self.features = [int(f) for f in features]
        self.architecture_type = architecture_type
        self.layer_norm = bool(layer_norm)
        self.observation_dim = tuple(int(d) for d in np.atleast_1d(observation_dim))
        self.learning_rate = learning_rate
        self.adam_eps = adam_eps
        self.gamma = gamma
        self.update_horizon = update_horizon
        self.precision = precision
        self.huber_delta = float(huber_delta)
        self.device = device
        self._seed = int(key) if not isinstance(key, torch.Generator) else int(key.initial_seed())
        self._engine = None
        self._graphed = None
        self._make_engine(batch_size, init=True)

    # ------------------------------------------------------------------ engine management
    def _make_engine(self, batch_size: int, init: bool = False) -> None:
        old = self._engine
        eng = QNetEngine(
            self.observation_dim, self.n_actions, self._n_heads, self.features, self.architecture_type,
            self.layer_norm, batch_size, gamma_n=self.gamma**self.update_horizon, learning_rate=self.learning_rate,
            adam_eps=self.adam_eps, precision=self.precision, device=self.device, huber_delta=self.huber_delta,
        )
        if init:
            eng.init_params(self._seed)
        else:  # a batch of another size arrived: same parameter layout, new workspace
            eng.params.copy_(old.params)
            eng.adam_m.copy_(old.adam_m)
            eng.adam_v.copy_(old.adam_v)
            eng.adam_count.copy_(old.adam_count)
            eng.losses_accum.copy_(old.losses_accum)
        self._engine = eng
        self._graphed = None  # captured against the old engine's buffers
        self.params = DeviceParams(eng, eng.params)
        self.optimizer_state = {"count": eng.adam_count, "mu": eng.adam_m, "nu": eng.adam_v}
        self._engine_changed(old)

    def _engine_changed(self, old) -> None:
        """Hook: a new engine replaced ``old`` (subclasses re-home extra device state)."""

    def _engine_for(self, batch_size: int) -> QNetEngine:
        if self._engine.batch_size != batch_size:
            self._make_engine(batch_size)
        return self._engine

    def _bind(self, params):
        """Accept the agent's own handle, another DeviceParams, or a Flax-layout pytree."""
        if params is None or params is self.params:
            return None
        if isinstance(params, DeviceParams):
            return params.tensor
        tree = params["params"] if "params" in params else params
        t = torch.empty_like(self._engine.params)
        self._engine.import_flax(tree, target=t)
        return t

    # ------------------------------------------------------------------ batches
    def _c_batch(self, eng: QNetEngine, samples):
        if hasattr(samples, "frame_ids"):  # DeviceBatch from the device replay
            return eng.make_batch(
                frames=samples.frames, frame_stride=samples.frame_stride, frame_ids=samples.frame_ids,
                action=samples.action, reward=samples.reward, terminal=samples.is_terminal,
            )
        dev = eng.device
        as_t = lambda x, dt: torch.as_tensor(np.asarray(x) if not torch.is_tensor(x) else x).to(dev).to(dt).contiguous()
        action = as_t(samples.action, torch.int32)
        reward = as_t(samples.reward, torch.float32)
        terminal = as_t(samples.is_terminal, torch.uint8)
        if self.architecture_type == "fc":
            st = as_t(samples.state, torch.float32).reshape(len(action), -1)
            nx = as_t(samples.next_state, torch.float32).reshape(len(action), -1)
            return eng.make_batch(state=st, next_state=nx, action=action, reward=reward, terminal=terminal)
        st = as_t(samples.state, torch.uint8)
        nx = as_t(samples.next_state, torch.uint8)
        B, h, w, stack = st.shape
        planes = torch.empty(2 * B * stack, h * w, dtype=torch.uint8, device=dev)
        ids = torch.empty(B, 2 * stack, dtype=torch.int32, device=dev)
        _hip.check(
            eng.lib.isdqn_replay_deinterleave(_hip.ptr(st), _hip.ptr(nx), h, w, stack, B, _hip.ptr(planes), _hip.ptr(ids), _hip.stream_ptr(dev))
        )
        return eng.make_batch(frames=planes, frame_stride=h * w, frame_ids=ids, action=action, reward=reward, terminal=terminal)

    @staticmethod
    def _batch_len(samples) -> int:
        return int(samples.action.shape[0])

    def _obs_to_device(self, state):
        eng = self._engine
        if self.architecture_type == "fc":
            obs = torch.as_tensor(np.asarray(state, dtype=np.float32)).reshape(1, -1).to(eng.device)
            return dict(obs=obs)
        s = np.asarray(state)
        h, w, stack = s.shape
        planes = np.ascontiguousarray(np.moveaxis(s, -1, 0)).reshape(stack, h * w)
        if planes.dtype != np.uint8:
            planes = planes.astype(np.uint8)
        fr = torch.from_numpy(planes).to(eng.device)
        ids = torch.arange(stack, dtype=torch.int32, device=eng.device)
        return dict(frames=fr, frame_stride=h * w, frame_ids=ids)

    def _q_row(self, params, state) -> torch.Tensor:
        """network.apply on one observation: device row of n_heads * n_actions values."""
        return self._engine.forward(n_rows=1, params=self._bind(params), **self._obs_to_device(state))

    def get_model(self):
        return {"params": self.params.to_flax()}
