"""DQN with the reference's surface (slimdqn/networks/dqn.py:11-101) on the HIP engine.

One head of ``n_actions`` outputs.  ``learn_on_batch(params, params_target, optimizer_state, batch)`` regresses
Q(s)[a] on  r + (1 - terminal) * gamma**n * max_a Q_target(s')  (dqn.py:74-88): the next states go through the TARGET
parameters -- a copy of the online parameters refreshed every ``target_update_frequency`` steps (dqn.py:46-54) -- and the
states through the online ones (C ABI: isdqn_net_learn_on_batch_target, the iS-DQN kernels with n_heads = 1).
Same observable differences as iSDQN (int seed, in-place device handles, device loss accumulation)."""
from __future__ import annotations

import numpy as np
import torch

from slimdqn.networks._agent import DeviceParams, EngineAgent
from slimdqn.networks.architectures.dqn import DQNNet


class DQN(EngineAgent):
    def __init__(
        self,
        key,
        observation_dim,
        n_actions,
        features: list,
        layer_norm: bool,
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
    ):
        self.use_graph = bool(use_graph)  # update_online_params on a device replay replays a captured step (networks/_agent.py)
        self.network = DQNNet([int(f) for f in features], architecture_type, n_actions, layer_norm, False)
        self.data_to_update = data_to_update
        self.target_update_frequency = target_update_frequency
        self.target_params = None
        self._init_engine_agent(key, observation_dim, n_actions, 1, features, layer_norm, architecture_type, learning_rate,
                                gamma, update_horizon, adam_eps, batch_size, precision, device)
        self.target_params = self.params.copy()  # dqn.py:34
        self.cumulated_loss = 0

    def _engine_changed(self, old) -> None:
        if self.target_params is not None:  # re-home the target copy on the new engine (same layout)
            self.target_params = DeviceParams(self._engine, self.target_params.tensor)

    # ------------------------------------------------------------------ dqn.py:40-57
    def update_online_params(self, step: int, replay_buffer):
        if step % self.data_to_update == 0:
            target = self._target_tensor(self.target_params)
            eng = self._engine_for(replay_buffer._batch_size) if hasattr(replay_buffer, "_batch_size") else self._engine
            g = self._graphed_update(replay_buffer, learn=lambda cb: eng.learn_on_batch_target(cb, target), key=target.data_ptr())
            if g is not None:  # (captured against the target BUFFER: update_target_params refreshes it in place, nothing is captured again)
                g.run()
                return
            batch_samples = replay_buffer.sample()
            self.params, self.optimizer_state, _ = self.learn_on_batch(self.params, self.target_params, self.optimizer_state, batch_samples)
            # `cumulated_loss += loss` (dqn.py:47) happens on the device inside the step

    def update_target_params(self, step: int):
        if step % self.target_update_frequency == 0:
            # `self.target_params = self.params.copy()` (dqn.py:50) as an in-place refresh of ONE persistent buffer: the captured
            # step reads the target through a fixed pointer, so a target update neither re-captures nor re-instantiates a graph
            self.target_params.tensor.copy_(self.params.tensor)
            eng = self._engine
            self.cumulated_loss = self.cumulated_loss + float(eng.losses_accum.cpu().numpy()[0])
            eng.losses_accum.zero_()
            logs = {"loss": self.cumulated_loss / (self.target_update_frequency / self.data_to_update)}
            self.cumulated_loss = 0
            return True, logs
        return False, {}

    # ------------------------------------------------------------------ dqn.py:59-88
    def _target_tensor(self, params_target) -> torch.Tensor:
        t = self._bind(params_target)
        return self._engine.params if t is None else t

    def learn_on_batch(self, params, params_target, optimizer_state, batch_samples):
        """In-place gradient step; returns (params, optimizer_state, loss) like the reference (loss: device scalar)."""
        eng = self._engine_for(self._batch_len(batch_samples))
        bound = self._bind(params)
        if bound is not None:
            eng.params.copy_(bound)
        losses = eng.learn_on_batch_target(self._c_batch(eng, batch_samples), self._target_tensor(params_target))
        return self.params, self.optimizer_state, losses[0]

    def loss_on_batch(self, params, params_target, samples):
        eng = self._engine_for(self._batch_len(samples))
        losses = eng.loss_on_batch_target(self._c_batch(eng, samples), self._target_tensor(params_target), params=self._bind(params))
        return losses[0]

    def compute_target(self, params, sample):
        """reward + (1 - terminal) * gamma**n * max_a Q_params(next_state) for ONE sample (dqn.py:84-88); host helper."""
        nq = self._q_row(params, sample.next_state).cpu().numpy().reshape(-1).astype(np.float64)
        return float(sample.reward) + (1.0 - float(sample.is_terminal)) * (self.gamma**self.update_horizon) * float(nq.max())

    def loss(self, params, params_target, sample):
        q = self._q_row(params, sample.state).cpu().numpy().reshape(-1).astype(np.float64)
        return float((q[int(sample.action)] - self.compute_target(params_target, sample)) ** 2)

    def q_values(self, params, state) -> np.ndarray:
        return self._q_row(params, state).cpu().numpy().reshape(self.n_actions)

    def best_action(self, params, state, **kwargs):
        return self._best_action(params, state, 0)

    def best_actions(self, params, states, **kwargs) -> np.ndarray:
        return self._best_actions(params, states, np.zeros(len(states), dtype=np.int32))
