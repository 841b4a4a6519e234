"""ORACLE (test infrastructure) -- the DQN and target-free DQN baselines, CPU restatement (torch CPU tensors).

PARITY UNPINNED for the network numerics (see oracle/network.py); the reference's tests/test_dqn.py and
tests/test_tfdqn.py hold formulas (target, loss, best action), restated as properties in tests/test_oracle_dqn.py.

  * DQN   follows slimdqn/networks/dqn.py:   __init__ :14-38 (target_params = params.copy()),
          update_online_params :40-47, update_target_params :49-57 (copy + loss log),
          learn_on_batch :59-72, loss_on_batch :74-76 (mean over the batch of the per-sample loss :78-82),
          compute_target :84-88 (next state through the TARGET parameters), best_action :90-93
  * TFDQN follows slimdqn/networks/tfdqn.py: update_* :38-54 (no target copy), learn_on_batch :56-64,
          loss_on_batch :66-80 (one forward on concat(state, next_state), stop-gradient target), compute_target :82-86;
          batch_norm=True (:62-63, 69-71, 91): training-mode forward with mutable batch_stats, learn_on_batch keeps the moved running
          averages, best_action uses them.  (DQN has no batch_norm argument: dqn.py:14-27, and dqn.py:86 applies the network
          without a mutable collection.)
"""
from __future__ import annotations

import numpy as np
import torch

from oracle import network as net
from oracle.isdqn import iSDQN as _Shared


class _OneHead:
    """Pieces both baselines share: parameters, Adam state, forward, acting."""

    def _init_common(self, key, observation_dim, n_actions, features, layer_norm, architecture_type, learning_rate, gamma,
                     update_horizon, data_to_update, target_update_frequency, adam_eps, dtype, params, batch_norm=False):
        self.batch_norm = bool(batch_norm)
        self.n_actions = n_actions
        self.features = [int(f) for f in features]
        self.architecture_type = architecture_type
        self.layer_norm = layer_norm
        self.dtype = dtype
        if params is None:
            params = net.init_params(int(key), observation_dim, self.features, architecture_type, n_actions, layer_norm, batch_norm=self.batch_norm)
        self.params = net.to_torch(params, dtype)
        self.batch_stats = net.to_torch(net.init_batch_stats(params), dtype) if self.batch_norm else None
        self._new_stats = None
        self.optimizer_state = {
            "count": 0,
            "mu": {m: {n: torch.zeros_like(t) for n, t in l.items()} for m, l in self.params.items()},
            "nu": {m: {n: torch.zeros_like(t) for n, t in l.items()} for m, l in self.params.items()},
        }
        self.learning_rate, self.adam_eps = learning_rate, adam_eps
        self.gamma, self.update_horizon = gamma, update_horizon
        self.data_to_update, self.target_update_frequency = data_to_update, target_update_frequency
        self.cumulated_loss = 0.0

    def apply(self, params, state, use_running_average: bool = False):
        state = torch.as_tensor(np.asarray(state)) if not torch.is_tensor(state) else state
        self._new_stats = {} if (self.batch_norm and not use_running_average) else None
        return net.forward(params, state, self.features, self.architecture_type, self.layer_norm, batch_norm=self.batch_norm,
                           batch_stats=self.batch_stats, use_running_average=use_running_average, new_stats=self._new_stats)  # (N, A)

    _batch_tensors = _Shared._batch_tensors

    def _adam(self, params, optimizer_state, grads):
        count = optimizer_state["count"] + 1
        b1, b2 = net.ADAM_B1, net.ADAM_B2
        c1, c2 = 1.0 - b1**count, 1.0 - b2**count
        new_p, new_mu, new_nu = {}, {}, {}
        for m, leaves in params.items():
            new_p[m], new_mu[m], new_nu[m] = {}, {}, {}
            for n, p in leaves.items():
                g = grads[m][n]
                mu = b1 * optimizer_state["mu"][m][n] + (1 - b1) * g
                nu = b2 * optimizer_state["nu"][m][n] + (1 - b2) * (g * g)
                new_p[m][n] = (p.detach() - self.learning_rate * (mu / c1) / (torch.sqrt(nu / c2) + self.adam_eps)).detach()
                new_mu[m][n], new_nu[m][n] = mu, nu
        return new_p, {"count": count, "mu": new_mu, "nu": new_nu}

    def _grads(self, params, loss_fn):
        leaves = [t for l in params.values() for t in l.values()]
        req = [t.detach().clone().requires_grad_(True) for t in leaves]
        it = iter(req)
        p2 = {m: {n: next(it) for n in l} for m, l in params.items()}
        loss = loss_fn(p2)
        g = torch.autograd.grad(loss, req)
        it = iter(g)
        return {m: {n: next(it) for n in l} for m, l in params.items()}, float(loss.detach())

    def best_action(self, params, state, **kwargs):
        return int(torch.argmax(self.apply(params, torch.as_tensor(np.asarray(state))[None], use_running_average=self.batch_norm)[0]))

    def get_model(self):
        # the reference pickles `{"params": self.params}` with self.params the full Flax variables dict (isdqn.py:137-138)
        if self.batch_norm:
            return {"params": {"params": net.to_numpy(self.params), "batch_stats": net.to_numpy(self.batch_stats)}}
        return {"params": {"params": net.to_numpy(self.params)}}


class DQN(_OneHead):
    def __init__(self, key, observation_dim, n_actions, features, layer_norm, architecture_type, learning_rate, gamma,
                 update_horizon, data_to_update, target_update_frequency, adam_eps=1e-8, dtype=torch.float32, params=None):
        self._init_common(key, observation_dim, n_actions, features, layer_norm, architecture_type, learning_rate, gamma,
                          update_horizon, data_to_update, target_update_frequency, adam_eps, dtype, params)
        self.target_params = {m: {n: t.clone() for n, t in l.items()} for m, l in self.params.items()}

    def update_online_params(self, step, replay_buffer):
        if step % self.data_to_update == 0:
            batch = replay_buffer.sample()
            self.params, self.optimizer_state, loss = self.learn_on_batch(self.params, self.target_params, self.optimizer_state, batch)
            self.cumulated_loss += loss

    def update_target_params(self, step):
        if step % self.target_update_frequency == 0:
            self.target_params = {m: {n: t.clone() for n, t in l.items()} for m, l in self.params.items()}
            logs = {"loss": self.cumulated_loss / (self.target_update_frequency / self.data_to_update)}
            self.cumulated_loss = 0.0
            return True, logs
        return False, {}

    def loss_terms(self, params, params_target, samples):
        state, action, reward, next_state, terminal = self._batch_tensors(samples)
        B = state.shape[0]
        q = self.apply(params, state).gather(1, action.view(B, 1)).squeeze(1)
        nq = self.apply(params_target, next_state).detach()
        targets = reward + (1 - terminal) * (self.gamma**self.update_horizon) * nq.max(dim=-1).values
        return q, targets, (q - targets) ** 2

    def loss_on_batch(self, params, params_target, samples):
        return self.loss_terms(params, params_target, samples)[2].mean()

    def grads(self, params, params_target, samples):
        return self._grads(params, lambda p: self.loss_on_batch(p, params_target, samples))

    def learn_on_batch(self, params, params_target, optimizer_state, samples):
        grads, loss = self.grads(params, params_target, samples)
        new_p, new_state = self._adam(params, optimizer_state, grads)
        return new_p, new_state, loss


class TFDQN(_OneHead):
    def __init__(self, key, observation_dim, n_actions, features, layer_norm, batch_norm, architecture_type, learning_rate,
                 gamma, update_horizon, data_to_update, target_update_frequency, adam_eps=1e-8, dtype=torch.float32, params=None):
        self._init_common(key, observation_dim, n_actions, features, layer_norm, architecture_type, learning_rate, gamma,
                          update_horizon, data_to_update, target_update_frequency, adam_eps, dtype, params, batch_norm=batch_norm)

    def update_online_params(self, step, replay_buffer):
        if step % self.data_to_update == 0:
            batch = replay_buffer.sample()
            self.params, self.optimizer_state, loss = self.learn_on_batch(self.params, self.optimizer_state, batch)
            self.cumulated_loss += loss

    def update_target_params(self, step):
        if step % self.target_update_frequency == 0:
            logs = {"loss": self.cumulated_loss / (self.target_update_frequency / self.data_to_update)}
            self.cumulated_loss = 0.0
            return True, logs
        return False, {}

    def loss_terms(self, params, samples):
        state, action, reward, next_state, terminal = self._batch_tensors(samples)
        B = state.shape[0]
        all_q = self.apply(params, torch.cat((state, next_state)))  # (2B, A)
        q = all_q[:B].gather(1, action.view(B, 1)).squeeze(1)
        targets = (reward + (1 - terminal) * (self.gamma**self.update_horizon) * all_q[B:].max(dim=-1).values).detach()
        return q, targets, (q - targets) ** 2

    def loss_on_batch(self, params, samples):
        return self.loss_terms(params, samples)[2].mean(), None

    def grads(self, params, samples):
        return self._grads(params, lambda p: self.loss_on_batch(p, samples)[0])

    def learn_on_batch(self, params, optimizer_state, samples):
        grads, loss = self.grads(params, samples)
        if self.batch_norm:  # params["batch_stats"] = batch_stats["batch_stats"] (tfdqn.py:62-63)
            self.batch_stats = {m: {n: t.detach().to(self.dtype) for n, t in l.items()} for m, l in self._new_stats.items()}
        new_p, new_state = self._adam(params, optimizer_state, grads)
        return new_p, new_state, loss
