"""Target-free DQN with the reference's surface (slimdqn/networks/tfdqn.py:11-101) on the HIP engine.

One head; the next states go through the SAME parameters as the states (one forward on concat(state, next_state),
stop-gradient target, tfdqn.py:68-86): the iS-DQN step with a single head that is regressed on its own target
(C ABI: isdqn_net_learn_on_batch with n_heads = 1).  ``update_target_params`` only reports the loss (tfdqn.py:47-54).
BatchNorm variants are outside the hot-path scope."""
from __future__ import annotations

import numpy as np

from slimdqn.networks._agent import EngineAgent
from slimdqn.networks.architectures.dqn import DQNNet


class TFDQN(EngineAgent):
    def __init__(
        self,
        key,
        observation_dim,
        n_actions,
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
    ):
        self.use_graph = bool(use_graph)  # update_online_params on a device replay replays a captured step (networks/_agent.py)
        self.network = DQNNet([int(f) for f in features], architecture_type, n_actions, layer_norm, batch_norm)
        self.data_to_update = data_to_update
        self.target_update_frequency = target_update_frequency
        self._init_engine_agent(key, observation_dim, n_actions, 1, features, layer_norm, architecture_type, learning_rate,
                                gamma, update_horizon, adam_eps, batch_size, precision, device, batch_norm=batch_norm)
        self.cumulated_loss = 0

    # ------------------------------------------------------------------ tfdqn.py:38-54
    def update_online_params(self, step: int, replay_buffer):
        if step % self.data_to_update == 0:
            g = self._graphed_update(replay_buffer)
            if g is not None:
                g.run()  # same draws, same kernels, same bits as the eager branch below
                return
            batch_samples = replay_buffer.sample()
            self.params, self.optimizer_state, _ = self.learn_on_batch(self.params, self.optimizer_state, batch_samples)

    def update_target_params(self, step: int):
        if step % self.target_update_frequency == 0:
            eng = self._engine
            self.cumulated_loss = self.cumulated_loss + float(eng.losses_accum.cpu().numpy()[0])
            eng.losses_accum.zero_()
            logs = {"loss": self.cumulated_loss / (self.target_update_frequency / self.data_to_update)}
            self.cumulated_loss = 0
            return True, logs
        return False, {}

    # ------------------------------------------------------------------ tfdqn.py:56-86
    def learn_on_batch(self, params, optimizer_state, batch_samples):
        eng = self._engine_for(self._batch_len(batch_samples))
        bound = self._bind(params)
        if bound is not None:
            eng.params.copy_(bound)
        losses = eng.learn_on_batch(self._c_batch(eng, batch_samples))
        return self.params, self.optimizer_state, losses[0]

    def loss_on_batch(self, params, samples):
        eng = self._engine_for(self._batch_len(samples))
        losses = eng.loss_on_batch(self._c_batch(eng, samples), params=self._bind(params))
        return losses[0], None

    def compute_target(self, sample, next_q_values):
        """reward + (1 - terminal) * gamma**n * max_a next_q (tfdqn.py:82-86); host helper."""
        nq = np.asarray(next_q_values, np.float64)
        r = np.asarray(sample.reward, np.float64)
        t = np.asarray(sample.is_terminal, np.float64)
        return r + (1 - t) * (self.gamma**self.update_horizon) * nq.max(-1)

    def q_values(self, params, state) -> np.ndarray:
        return self._q_row(params, state).cpu().numpy().reshape(self.n_actions)

    def best_action(self, params, state, **kwargs):
        return self._best_action(params, state, 0)

    def best_actions(self, params, states, **kwargs) -> np.ndarray:
        return self._best_actions(params, states, np.zeros(len(states), dtype=np.int32))
