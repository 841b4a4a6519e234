"""ORACLE (test infrastructure) -- the iS-DQN agent, CPU restatement (torch CPU tensors).

PARITY UNPINNED for the network numerics (see oracle/network.py).  Follows the
reference ``slimdqn/networks/isdqn.py``:

  * __init__ / apply wrapper ... :14-53  (final feature (1+K)*A viewed as (-1, 1+K, A),
                                          column  k*A + a ;  last_idx_mlp :33)
  * update_online_params ....... :55-62
  * update_target_params ....... :64-80  (log normaliser T / data_to_update)
  * learn_on_batch ............. :82-90
  * loss_on_batch .............. :92-103 (one forward on concat(state, next_state);
                                          q = heads 1..K at the taken action, rows [:B];
                                          targets from heads 0..K-1, rows [B:];
                                          squared TD error, mean over batch, sum over heads)
  * compute_target ............. :105-109 (r + (1-terminal) * gamma**n * max_a)
  * shift_params ............... :111-125 (head k <- head k+1, moments untouched)
  * best_action ................ :127-135 (head 1+idx)
  * batch_norm=True (cnn / fc) . :40, 87-88, 95, 130: loss / learn run the network in training mode on concat(state, next_state)
                                  (batch statistics; the gradient flows through them into BOTH halves), learn_on_batch keeps the
                                  moved running averages (``self.batch_stats``: Flax's second collection, kept beside the
                                  parameter dict here), best_action uses them (use_running_average=True)
"""
from __future__ import annotations

import numpy as np
import torch

from oracle import network as net


class iSDQN:
    def __init__(
        self,
        key,  # int seed (JAX threefry keys are not reproducible offline)
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
        dtype=torch.float32,
        params=None,
        huber_delta: float = 0.0,
    ):
        """``huber_delta``: 0 = the reference's squared TD error (isdqn.py:102); > 0 = Huber loss (optax.huber_loss: 0.5 d^2 for
        |d| <= delta, delta (|d| - delta / 2) beyond) -- the north star's wording, not in the reference."""
        self.huber_delta = float(huber_delta)
        self.batch_norm = bool(batch_norm)
        self.n_bellman_iterations = n_bellman_iterations
        self.n_actions = n_actions
        self.features = [int(f) for f in features]
        self.architecture_type = architecture_type
        self.layer_norm = layer_norm
        self.last_idx_mlp = len(features) if architecture_type == "fc" else len(features) - 3  # (cnn and impala: isdqn.py:33)
        self.final_feature = (1 + n_bellman_iterations) * n_actions
        self.dtype = dtype
        if params is None:
            params = net.init_params(
                int(key), observation_dim, self.features, architecture_type, self.final_feature, layer_norm, batch_norm=self.batch_norm
            )
        self.params = net.to_torch(params, dtype)
        self.batch_stats = net.to_torch(net.init_batch_stats(params), dtype) if self.batch_norm else None
        self._new_stats = None
        self.optimizer_state = {
            "count": 0,
            "mu": {m: {n: torch.zeros_like(t) for n, t in l.items()} for m, l in self.params.items()},
            "nu": {m: {n: torch.zeros_like(t) for n, t in l.items()} for m, l in self.params.items()},
        }
        self.learning_rate = learning_rate
        self.adam_eps = adam_eps
        self.gamma = gamma
        self.update_horizon = update_horizon
        self.data_to_update = data_to_update
        self.target_update_frequency = target_update_frequency
        self.cumulated_losses = np.zeros(self.n_bellman_iterations)

    # -- network ---------------------------------------------------------------
    def apply(self, params, state, use_running_average: bool = False):
        """(N, 1+K, A) head view of the network output.  BatchNorm: training mode unless ``use_running_average`` (the moved
        running averages of a training-mode call are left in ``self._new_stats``: isdqn.py:40 returns them as batch_stats)."""
        state = torch.as_tensor(np.asarray(state)) if not torch.is_tensor(state) else state
        self._new_stats = {} if (self.batch_norm and not use_running_average) else None
        q = net.forward(params, state, self.features, self.architecture_type, self.layer_norm, batch_norm=self.batch_norm,
                        batch_stats=self.batch_stats, use_running_average=use_running_average, new_stats=self._new_stats)
        return q.reshape(-1, 1 + self.n_bellman_iterations, self.n_actions)

    # -- trainer-facing cadence -------------------------------------------------
    def update_online_params(self, step: int, replay_buffer):
        if step % self.data_to_update == 0:
            batch = replay_buffer.sample()
            self.params, self.optimizer_state, losses = self.learn_on_batch(self.params, self.optimizer_state, batch)
            self.cumulated_losses += losses

    def update_target_params(self, step: int):
        if step % self.target_update_frequency == 0:
            self.params = self.shift_params(self.params)
            norm = self.target_update_frequency / self.data_to_update
            logs = {"loss": np.mean(self.cumulated_losses) / norm}
            for k in range(min(self.n_bellman_iterations, 5)):
                logs[f"networks/{k}_loss"] = self.cumulated_losses[k] / norm
            self.cumulated_losses = np.zeros_like(self.cumulated_losses)
            return True, logs
        return False, {}

    # -- loss / target -------------------------------------------------------------
    def _batch_tensors(self, samples):
        action = torch.as_tensor(np.asarray(samples.action)).long()
        reward = torch.as_tensor(np.asarray(samples.reward)).to(self.dtype)
        terminal = torch.as_tensor(np.asarray(samples.is_terminal)).to(self.dtype)
        state = torch.as_tensor(np.asarray(samples.state))
        next_state = torch.as_tensor(np.asarray(samples.next_state))
        return state, action, reward, next_state, terminal

    def compute_target(self, reward, is_terminal, next_q_values):
        """next_q_values (..., A) -> reward + (1-terminal) * gamma**n * max_a."""
        return reward + (1 - is_terminal) * (self.gamma**self.update_horizon) * next_q_values.max(dim=-1).values

    def loss_terms(self, params, samples):
        """(q_values (B,K), targets (B,K), td (B,K)) -- the pieces of loss_on_batch."""
        state, action, reward, next_state, terminal = self._batch_tensors(samples)
        B = state.shape[0]
        all_q = self.apply(params, torch.cat((state, next_state)))  # (2B, 1+K, A)
        q_values = all_q[:B, 1:].gather(2, action.view(B, 1, 1).expand(B, self.n_bellman_iterations, 1)).squeeze(2)
        targets = self.compute_target(reward[:, None], terminal[:, None], all_q[B:, :-1]).detach()
        d = q_values - targets
        if self.huber_delta > 0:
            a = d.abs()
            td = torch.where(a <= self.huber_delta, 0.5 * d * d, self.huber_delta * (a - 0.5 * self.huber_delta))
        else:
            td = d**2
        return q_values, targets, td

    def loss_on_batch(self, params, samples):
        _q, _t, td = self.loss_terms(params, samples)
        per_head = td.mean(dim=0)
        return per_head.sum(), (per_head, None)

    # -- gradient step -------------------------------------------------------------
    def grads(self, params, samples):
        leaves = [t for l in params.values() for t in l.values()]
        req = [t.detach().clone().requires_grad_(True) for t in leaves]
        it = iter(req)
        p2 = {m: {n: next(it) for n in l} for m, l in params.items()}
        loss, (per_head, _) = self.loss_on_batch(p2, samples)
        g = torch.autograd.grad(loss, req)
        it = iter(g)
        grads = {m: {n: next(it) for n in l} for m, l in params.items()}
        return grads, per_head.detach()

    def learn_on_batch(self, params, optimizer_state, samples):
        grads, per_head = self.grads(params, samples)
        if self.batch_norm:  # params["batch_stats"] = batch_stats["batch_stats"] (isdqn.py:87-88)
            self.batch_stats = {m: {n: t.detach().to(self.dtype) for n, t in l.items()} for m, l in self._new_stats.items()}
        count = optimizer_state["count"] + 1
        b1, b2 = net.ADAM_B1, net.ADAM_B2
        c1 = 1.0 - b1**count
        c2 = 1.0 - b2**count
        new_p, new_mu, new_nu = {}, {}, {}
        for m, leaves in params.items():
            new_p[m], new_mu[m], new_nu[m] = {}, {}, {}
            for n, p in leaves.items():
                g = grads[m][n]
                mu = b1 * optimizer_state["mu"][m][n] + (1 - b1) * g
                nu = b2 * optimizer_state["nu"][m][n] + (1 - b2) * (g * g)
                update = (mu / c1) / (torch.sqrt(nu / c2) + self.adam_eps)
                new_p[m][n] = (p.detach() - self.learning_rate * update).detach()
                new_mu[m][n], new_nu[m][n] = mu, nu
        return new_p, {"count": count, "mu": new_mu, "nu": new_nu}, per_head.numpy().astype(np.float64)

    # -- head shift / acting -------------------------------------------------------
    def shift_params(self, params):
        A = self.n_actions
        name = f"Dense_{self.last_idx_mlp}"
        out = {m: dict(l) for m, l in params.items()}
        kernel = params[name]["kernel"].clone()
        kernel[:, :-A] = params[name]["kernel"][:, A:]
        bias = params[name]["bias"].clone()
        bias[:-A] = params[name]["bias"][A:]
        out[name] = {"kernel": kernel, "bias": bias}
        return out

    def best_action(self, params, state, idx_network: int):
        """argmax of online head ``1 + idx_network`` for one state (isdqn.py:127-135).

        The reference draws idx_network with jax.random.randint(key, (), 0, K); the
        draw is an input here because threefry is not reproducible offline.
        """
        state = torch.as_tensor(np.asarray(state))
        q = self.apply(params, state[None], use_running_average=True)[0]
        return int(torch.argmax(q[1 + idx_network]))

    def get_model(self):
        # the reference pickles `{"params": self.params}` with self.params the full Flax variables dict (isdqn.py:137-138)
        if self.batch_norm:
            return {"params": {"params": net.to_numpy(self.params), "batch_stats": net.to_numpy(self.batch_stats)}}
        return {"params": {"params": net.to_numpy(self.params)}}
