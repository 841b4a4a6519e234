"""iterated Shared DQN agent with the reference's surface (slimdqn/networks/isdqn.py:13-138), running
on the HIP engine (csrc/net_kernels.hip) through the C ABI.

One network whose last Dense has (1+K)*A outputs viewed as heads [Q_0 .. Q_K] (isdqn.py:34-41); head k
is regressed onto the Bellman target built from head k-1 of the same parameters (isdqn.py:92-109);
``update_target_params`` shifts the heads by one slot (isdqn.py:111-125).

Differences a caller can observe (all recorded in DESIGN.md):
  * ``key`` is an int seed (JAX threefry streams cannot be reproduced without JAX);
  * ``params`` / ``optimizer_state`` are opaque device handles -- the update runs in place and the
    methods return the same handles, which is equivalent because every reference caller rebinds them
    (isdqn.py:59-61); ``get_model()`` still returns the reference's ``{"params": pytree}`` with Flax
    names, layouts and shapes;
  * batches are ``DeviceBatch`` objects from the device replay (reference-layout batches with
    ``.state`` arrays of shape (B,h,w,stack) are accepted too and de-interleaved on the device);
  * ``cumulated_losses`` accumulates on the device and is read back only at target updates.
"""
from __future__ import annotations

import numpy as np
import torch

from slimdqn._engine import QNetEngine
from slimdqn.networks._agent import DeviceParams, EngineAgent
from slimdqn.networks.architectures.dqn import DQNNet


class iSDQN(EngineAgent):
    def __init__(
        self,
        key,
        observation_dim,
        n_actions,
        n_bellman_iterations: int,
        features: list,
        layer_norm: bool,
        batch_norm: bool,
        architecture_type: str,
        learning_rate: float,
        gamma: float,
        update_horizon: int,
        data_to_update: int,
        target_update_frequency: int,
        adam_eps: float = 1e-8,
        batch_size: int = 32,
        precision: str = "bf16x3",
        device: str | None = None,
        use_graph: bool = True,
        huber_delta: float = 0.0,
    ):
        """``huber_delta``: 0 keeps the reference's squared TD error (isdqn.py:102); > 0 trains on the Huber loss."""
        self.n_bellman_iterations = n_bellman_iterations
        self.last_idx_mlp = len(features) if architecture_type == "fc" else len(features) - 3
        self.network = DQNNet([int(f) for f in features], architecture_type, (1 + n_bellman_iterations) * n_actions, layer_norm, batch_norm)
        self.data_to_update = data_to_update
        self.target_update_frequency = target_update_frequency
        # update_online_params on a device replay runs as a captured single step (hipGraph: one launch instead of ~25
        # kernel launches on two streams); priority_writeback adds the sum-tree write-back of the step's TD errors
        self.use_graph = bool(use_graph)
        self.priority_writeback = False
        self._init_engine_agent(key, observation_dim, n_actions, 1 + n_bellman_iterations, features, layer_norm, architecture_type,
                                learning_rate, gamma, update_horizon, adam_eps, batch_size, precision, device, huber_delta, batch_norm)
        self._action_rng = np.random.default_rng(self._seed + 1)
        self.cumulated_losses = np.zeros(self.n_bellman_iterations)

    # ------------------------------------------------------------------ isdqn.py:55-80
    def learn_steps(self, n_steps: int, replay_buffer) -> None:
        """``n_steps`` consecutive gradient steps (sample -> learn -> [write-back] each) on an unchanged replay buffer: what
        n_steps calls of ``update_online_params`` at update steps do, as ONE graph replay when the replay is this GPU's
        device replay (a round of vectorised environments owes several steps at once; per-step replays cost the host
        ~280 us each -- about what the GPU needs for the step)."""
        if n_steps <= 0:
            return
        # ONE executable graph per agent (networks/_agent.py): the round's step count selects what is captured.  A caller that keeps
        # changing it (a handful of captures) gets replays of whatever is live, or single-step replays, instead of a capture per call.
        live, settled = self._graphed, getattr(self, "_captures", 0) >= 8
        if live is not None and live.rb is replay_buffer and n_steps % live.S == 0 and (live.S == n_steps or settled):
            steps = live.S
        else:
            steps = 1 if settled else n_steps
        g = self._graphed_update(replay_buffer, steps=steps)
        if g is None:
            for _ in range(n_steps):
                self.update_online_params(0, replay_buffer)
            return
        for _ in range(n_steps // g.S):
            g.run()

    def update_online_params(self, step: int, replay_buffer):
        if step % self.data_to_update == 0:
            live = self._graphed
            if live is not None and live.S > 1 and live.rb is replay_buffer:  # learn_steps' capture is the live graph: eager step
                g = None
            else:
                g = self._graphed_update(replay_buffer)
            if g is not None:
                g.run()  # same draws, same kernels, same bits as the eager branch below (tests/test_gpu_graphed_update.py)
                return
            batch_samples = replay_buffer.sample()
            self.params, self.optimizer_state, _ = self.learn_on_batch(self.params, self.optimizer_state, batch_samples)
            if self.priority_writeback and hasattr(replay_buffer, "update_device"):
                replay_buffer.update_device(batch_samples, self._engine.priorities)
            # `cumulated_losses += losses` (isdqn.py:62) happens on the device inside the step

    def update_target_params(self, step: int):
        if step % self.target_update_frequency == 0:
            self.params = self.shift_params(self.params)
            eng = self._engine
            self.cumulated_losses = self.cumulated_losses + eng.losses_accum.cpu().numpy().astype(np.float64)
            eng.losses_accum.zero_()
            norm = self.target_update_frequency / self.data_to_update
            logs = {"loss": np.mean(self.cumulated_losses) / norm}
            for idx_network in range(min(self.n_bellman_iterations, 5)):
                logs[f"networks/{idx_network}_loss"] = self.cumulated_losses[idx_network] / norm
            self.cumulated_losses = np.zeros_like(self.cumulated_losses)
            return True, logs
        return False, {}

    # ------------------------------------------------------------------ isdqn.py:82-109
    def learn_on_batch(self, params, optimizer_state, batch_samples):
        """In-place gradient step; returns (params, optimizer_state, losses[K]) like the reference.
        ``losses`` is a device tensor (reading it synchronises)."""
        eng = self._engine_for(self._batch_len(batch_samples))
        bound = self._bind(params)
        if bound is not None:
            eng.params.copy_(bound)
        losses = eng.learn_on_batch(self._c_batch(eng, batch_samples))
        return self.params, self.optimizer_state, losses

    def loss_on_batch(self, params, samples):
        eng = self._engine_for(self._batch_len(samples))
        losses = eng.loss_on_batch(self._c_batch(eng, samples), params=self._bind(params))
        return losses.sum(), (losses, None)

    def compute_target(self, sample, next_q_values):
        """reward + (1 - terminal) * gamma**n * max_a next_q  (isdqn.py:105-109); host-level helper."""
        nq = torch.as_tensor(np.asarray(next_q_values) if not torch.is_tensor(next_q_values) else next_q_values)
        r = torch.as_tensor(np.asarray(sample.reward, dtype=np.float32))
        t = torch.as_tensor(np.asarray(sample.is_terminal, dtype=np.float32))
        return r + (1 - t) * (self.gamma**self.update_horizon) * nq.max(dim=-1).values.cpu()

    # ------------------------------------------------------------------ isdqn.py:111-138
    def shift_params(self, params):
        t = self._bind(params)
        self._engine.shift_params(params=t)
        return self.params if t is None else DeviceParams(self._engine, t)

    def q_values(self, params, state) -> np.ndarray:
        """(1+K, A) head view for one observation -- ``network.apply(...).reshape`` of isdqn.py:130-132."""
        return self._q_row(params, state).cpu().numpy().reshape(1 + self.n_bellman_iterations, self.n_actions)

    def best_action(self, params, state, key=None):
        """argmax of a uniformly drawn online head (isdqn.py:127-135).  ``key``: None (agent's own
        generator), an int head index, or a numpy Generator."""
        if key is None:
            idx = int(self._action_rng.integers(0, self.n_bellman_iterations))
        elif isinstance(key, (int, np.integer)):
            idx = int(key)
        else:
            idx = int(key.integers(0, self.n_bellman_iterations))
        return self._best_action(params, state, idx)

    def best_actions_planes(self, params, planes, rows, key=None, wait: bool = True):
        """``best_actions`` on the planar host block of a VectorEnv (environments/vector.py) for its environments ``rows``.
        ``wait=False`` returns a function that waits for THIS forward and returns the actions (the caller enqueues more work first)."""
        n = len(rows)
        if key is None:
            idx = self._action_rng.integers(0, self.n_bellman_iterations, size=n)
        elif isinstance(key, np.random.Generator):
            idx = key.integers(0, self.n_bellman_iterations, size=n)
        else:
            idx = np.broadcast_to(np.asarray(key), (n,))
        return self._best_actions_planes(params, planes, rows, idx, wait=wait)

    def best_actions(self, params, states, key=None) -> np.ndarray:
        """``best_action`` for n observations (vectorised host environments): one head draw per observation from ``key``
        (None: the agent's generator; an int array: the heads themselves), one forward, one read back."""
        n = len(states)
        if key is None:
            idx = self._action_rng.integers(0, self.n_bellman_iterations, size=n)
        elif isinstance(key, np.random.Generator):
            idx = key.integers(0, self.n_bellman_iterations, size=n)
        else:
            idx = np.broadcast_to(np.asarray(key), (n,))
        return self._best_actions(params, states, idx)
