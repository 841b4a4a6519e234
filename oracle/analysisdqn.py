"""ORACLE (test infrastructure) -- AnalysisDQN, CPU restatement (torch CPU tensors).  PARITY UNPINNED for the network
numerics (oracle/network.py).  Follows the reference ``slimdqn/networks/analysisdqn.py``:

  * __init__ (target_params = params.copy(), cumulated diagnostics) ....... :14-61
  * update_online_params (TWO batches per update: train, then eval) ........ :63-84
  * update_target_params (target copy BEFORE the head shift, log names) .... :86-121
  * learn_on_batch (targets of both batches before / after the update) ..... :123-160
      - batch_norm: every apply is a training-mode forward (mutable batch_stats, :40-42); the collection stored with the updated
        parameters is the evaluation batch's pre-update forward's (:121, :130-131)
  * grad_and_loss_on_batch ................................................ :162-219
      - loss_tb: head 1 of the states (params) on head 1 of the next states through params_target
      - loss_tf: head 1 on head 1 of the same parameters (stop-gradient)
      - loss_is: the iS-DQN loss (its gradient is the one the optimizer applies)
      - extract_feature_gradients: last Dense restricted to columns / entries A .. 2A (head 1), every leaf whose path contains
        "norm" dropped, the rest concatenated in sorted-path order; cosine(a, b) = a.b / (|a| |b| + 1e-9)
"""
from __future__ import annotations

import numpy as np
import torch

from oracle import network as net
from oracle.isdqn import iSDQN


def _detached(stats, dtype):
    return {m: {n: t.detach().to(dtype) for n, t in l.items()} for m, l in stats.items()}


class AnalysisDQN(iSDQN):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.target_params = {m: {n: t.clone() for n, t in l.items()} for m, l in self.params.items()}
        K = self.n_bellman_iterations
        self.cumulated_target_churns_train = np.zeros(K)
        self.cumulated_target_churns_eval = np.zeros(K)
        self.cumulated_cosine_sim_is_to_tb = 0.0
        self.cumulated_cosine_sim_tf_to_tb = 0.0

    def update_online_params(self, step: int, replay_buffer):
        if step % self.data_to_update == 0:
            batch = replay_buffer.sample()
            batch_eval = replay_buffer.sample()
            self.params, self.optimizer_state, losses, ct, ce, c_is, c_tf = self.learn_on_batch(
                self.params, self.target_params, self.optimizer_state, batch, batch_eval)
            self.cumulated_losses += losses
            self.cumulated_target_churns_train += ct
            self.cumulated_target_churns_eval += ce
            self.cumulated_cosine_sim_is_to_tb += c_is
            self.cumulated_cosine_sim_tf_to_tb += c_tf

    def update_target_params(self, step: int):
        if step % self.target_update_frequency == 0:
            self.target_params = {m: {n: t.clone() for n, t in l.items()} for m, l in self.params.items()}
            self.params = self.shift_params(self.params)
            norm = self.target_update_frequency / self.data_to_update
            logs = {
                "loss": np.mean(self.cumulated_losses) / norm,
                "analysis/target_churns_train": self.cumulated_target_churns_train[0] / norm,
                "analysis/target_churns_eval": self.cumulated_target_churns_eval[0] / norm,
                "analysis/cosine_sim_iS_to_TB": self.cumulated_cosine_sim_is_to_tb / norm,
                "analysis/cosine_sim_TF_to_TB": self.cumulated_cosine_sim_tf_to_tb / norm,
            }
            for k in range(min(self.n_bellman_iterations, 5)):
                logs[f"networks/{k}_loss"] = self.cumulated_losses[k] / norm
                logs[f"networks/{k}_target_churns_train"] = self.cumulated_target_churns_train[k] / norm
                logs[f"networks/{k}_target_churns_eval"] = self.cumulated_target_churns_eval[k] / norm
            self.cumulated_losses = np.zeros_like(self.cumulated_losses)
            self.cumulated_target_churns_train = np.zeros_like(self.cumulated_target_churns_train)
            self.cumulated_target_churns_eval = np.zeros_like(self.cumulated_target_churns_eval)
            self.cumulated_cosine_sim_is_to_tb = 0.0
            self.cumulated_cosine_sim_tf_to_tb = 0.0
            return True, logs
        return False, {}

    # -- analysisdqn.py:162-219 -------------------------------------------------------------------------------------
    def _targets(self, params, samples):
        state, action, reward, next_state, terminal = self._batch_tensors(samples)
        B = state.shape[0]
        all_q = self.apply(params, torch.cat((state, next_state)))
        return self.compute_target(reward[:, None], terminal[:, None], all_q[B:, :-1]).detach()

    def _grad(self, loss_fn, params):
        leaves = [t for l in params.values() for t in l.values()]
        req = [t.detach().clone().requires_grad_(True) for t in leaves]
        it = iter(req)
        p2 = {m: {n: next(it) for n in l} for m, l in params.items()}
        g = torch.autograd.grad(loss_fn(p2), req)
        it = iter(g)
        return {m: {n: next(it) for n in l} for m, l in params.items()}

    def feature_gradient(self, grads):
        A, name = self.n_actions, f"Dense_{self.last_idx_mlp}"
        flat = {}
        for m, l in grads.items():
            for n, g in l.items():
                if m == name:
                    g = g[:, A : 2 * A] if n == "kernel" else g[A : 2 * A]
                flat[f"params/{m}/{n}"] = g
        keep = [flat[k].reshape(-1) for k in sorted(flat) if "norm" not in k.lower()]
        return torch.cat(keep)

    def three_gradients(self, params, params_target, samples):
        state, action, reward, next_state, terminal = self._batch_tensors(samples)
        B = state.shape[0]
        take = lambda q: q.gather(1, action.view(B, 1)).squeeze(1)

        def loss_tb(p):
            q = self.apply(p, state)[:, 1]
            nq = self.apply(params_target, next_state)[:, 1]
            return ((take(q) - self.compute_target(reward, terminal, nq).detach()) ** 2).mean()

        def loss_tf(p):
            all_q = self.apply(p, torch.cat((state, next_state)))
            return ((take(all_q[:B, 1]) - self.compute_target(reward, terminal, all_q[B:, 1]).detach()) ** 2).mean()

        def loss_is(p):
            return self.loss_on_batch(p, samples)[0]

        return self._grad(loss_is, params), self._grad(loss_tf, params), self._grad(loss_tb, params)

    @staticmethod
    def cosine(a, b) -> float:
        return float(torch.dot(a, b) / (torch.linalg.norm(a) * torch.linalg.norm(b) + 1e-9))

    def learn_on_batch(self, params, params_target, optimizer_state, batch, batch_eval):
        g_is, g_tf, g_tb = self.three_gradients(params, params_target, batch)
        f_is, f_tf, f_tb = (self.feature_gradient(g) for g in (g_is, g_tf, g_tb))
        train_pre, eval_pre = self._targets(params, batch), self._targets(params, batch_eval)
        # batch_norm: the collection stored with the new parameters is the one the LAST forward before the update returned -- the
        # evaluation batch's (analysisdqn.py:121 rebinds `batch_stats`, :130-131 stores it), not the training forward's
        eval_stats = _detached(self._new_stats, self.dtype) if self.batch_norm else None
        new_params, new_state, losses = iSDQN.learn_on_batch(self, params, optimizer_state, batch)
        if self.batch_norm:
            self.batch_stats = eval_stats
        train_post, eval_post = self._targets(new_params, batch), self._targets(new_params, batch_eval)
        return (new_params, new_state, losses, (train_pre - train_post).abs().mean(0).numpy().astype(np.float64),
                (eval_pre - eval_post).abs().mean(0).numpy().astype(np.float64), self.cosine(f_is, f_tb), self.cosine(f_tf, f_tb))


class AnalysisTFDQN:
    """slimdqn/networks/analysistfdqn.py:81-118 on oracle.dqn.TFDQN: the step plus the two target churns (scalars)."""

    def __init__(self, *args, **kwargs):
        from oracle.dqn import TFDQN

        self.agent = TFDQN(*args, **kwargs)

    def learn_on_batch(self, params, optimizer_state, batch, batch_eval):
        a = self.agent
        targets = lambda p, s: a.loss_terms(p, s)[1].detach()
        train_pre, eval_pre = targets(params, batch), targets(params, batch_eval)
        eval_stats = _detached(a._new_stats, a.dtype) if a.batch_norm else None  # (analysistfdqn.py:85-95: the evaluation forward's collection)
        new_params, new_state, loss = a.learn_on_batch(params, optimizer_state, batch)
        if a.batch_norm:
            a.batch_stats = eval_stats
        return (new_params, new_state, loss, float((train_pre - targets(new_params, batch)).abs().mean()),
                float((eval_pre - targets(new_params, batch_eval)).abs().mean()))
